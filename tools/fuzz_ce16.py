"""Randomised check of the split-fp16 logits pair (tt_ce16_fwd_du_keep / tt_ce16_bwd_kept, csrc/ce_f16x2.hip) against
float64 and next to the fp32-MFMA pair on the same inputs: random M (multiples of 256), N (multiples of 1024), diagonal
offset, operand scale 1e-3 ... 30, per-row magnitude spread up to 2^10, zero rows, zero / tiny dL/dce entries, one
all-zero operand now and then.  Criterion per output (ce, unit user gradient, item gradient):
error <= 4 x the fp32 pair's error + 2e-6 x max|reference| (16 x at operand scale 30, where the logits are ~1e4, their own
fp32 resolution is 1e-3 and the fp32 pair itself is 1e-4 ... 1e-3 off in relative terms: the split logits' 1.7 x larger
rounding error is amplified by exp() there); lse and the kept logits absolutely.
        python tools/fuzz_ce16.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N
from two_tower_models_amd import ops

lib = N.load()
DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
D = 128
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    M = 256 * int(rng.integers(1, 9))
    Nn = 1024 * int(rng.integers(max(1, (M + 1023) // 1024), 13))
    off = int(rng.integers(0, Nn - M + 1))
    scale = float(rng.choice([1e-3, 0.1, 0.35, 0.5, 3.0, 30.0]))
    spread = int(rng.choice([0, 0, 3, 10]))
    zero_op = int(rng.integers(0, 40)) == 0
    what = f"case {n}: M={M} N={Nn} off={off} scale={scale} spread=2^{spread} zero_operand={zero_op}"
    try:
        g = torch.Generator().manual_seed(1000 + n)
        U = torch.randn(M, D, generator=g) * scale
        I = torch.randn(Nn, D, generator=g) * scale
        if spread:
            U *= torch.exp2(-torch.rand(M, 1, generator=g) * spread)
            I *= torch.exp2(-torch.rand(Nn, 1, generator=g) * spread)
        U[torch.rand(M, generator=g) < 0.01] = 0.0
        I[torch.rand(Nn, generator=g) < 0.01] = 0.0
        if zero_op:
            U.zero_()
        coef = torch.rand(M, generator=g) / M * float(rng.choice([1.0, 1.0, 1e3, 1e-3, -5.0]))
        coef[torch.rand(M, generator=g) < 0.05] = 0.0
        coef[torch.rand(M, generator=g) < 0.05] *= 1e-6
        Ug, Ig, cg = U.to(DEV), I.to(DEV), coef.to(DEV)
        e = lambda *s: torch.full(s, float("nan"), device=DEV)
        wsn = lib.tt_ce16_workspace_bytes(M, Nn, D)
        ws = torch.empty(wsn, dtype=torch.uint8, device=DEV)
        lse, ce, du, Z, dI = e(M), e(M), e(M, D), e(M * Nn), e(Nn, D)
        N.check(lib.tt_ce16_fwd_du_keep(Ug.data_ptr(), D, Ig.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), ce.data_ptr(), du.data_ptr(), D,
                                        Z.data_ptr(), M * Nn * 4, ws.data_ptr(), wsn, N.stream()), "ce16 fwd")
        N.check(lib.tt_ce16_bwd_kept(Ug.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), cg.data_ptr(), Z.data_ptr(), M * Nn * 4,
                                     dI.data_ptr(), D, ws.data_ptr(), wsn, N.stream()), "ce16 bwd")
        dIr = e(Nn, D)  # the recomputing backward (no logits buffer), alternately trusting / re-forming the workspace images
        N.check(lib.tt_ce16_fwd_du_keep(Ug.data_ptr(), D, Ig.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), ce.data_ptr(), du.data_ptr(), D,
                                        None, 0, ws.data_ptr(), wsn, N.stream()), "ce16 fwd (no logits)")
        N.check(lib.tt_ce16_bwd_recompute(Ug.data_ptr(), D, Ig.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), cg.data_ptr(), dIr.data_ptr(), D,
                                          ws.data_ptr(), wsn, n % 2, N.stream()), "ce16 bwd recompute")
        wsp, wsn32 = ops._ws(DEV, lib.tt_inbatch_ce_workspace_bytes(M, Nn, D), "fuzz")
        lse32, ce32, du32, dI32 = e(M), e(M), e(M, D), e(Nn, D)
        zn = lib.tt_inbatch_ce_logits_bytes(M, Nn)
        Z32 = torch.empty(zn // 4, device=DEV)
        N.check(lib.tt_inbatch_ce_fwd_du_keep(Ug.data_ptr(), D, Ig.data_ptr(), D, M, Nn, D, off, lse32.data_ptr(), ce32.data_ptr(),
                                              du32.data_ptr(), D, Z32.data_ptr(), zn, wsp, wsn32, N.stream()), "fwd_du_keep")
        N.check(lib.tt_inbatch_ce_bwd_kept(Ug.data_ptr(), D, M, Nn, D, off, lse32.data_ptr(), cg.data_ptr(), Z32.data_ptr(), zn,
                                           dI32.data_ptr(), D, wsp, wsn32, N.stream()), "bwd_kept")
        Ud, Id, cd = Ug.double(), Ig.double(), cg.double()  # float64 on the GPU
        S = Ud @ Id.t()
        rows = torch.arange(M, device=DEV)
        ref_lse = torch.logsumexp(S, dim=1)
        ref_ce = ref_lse - S[rows, rows + off]
        P = torch.softmax(S, dim=1)
        ref_du = P @ Id - Id[rows + off]
        P[rows, rows + off] -= 1.0
        ref_dI = (P * cd.unsqueeze(1)).t() @ Ud
        err = lambda got, ref: float((got.double() - ref).abs().max())
        msgs = []
        smax = max(1.0, float(S.abs().max()))
        Zrm = Z.view(M // 32, Nn // 32, 32, 32).permute(0, 2, 1, 3).reshape(M, Nn)
        if err(Zrm, S * 1.4426950408889634) > 2e-6 * smax:
            msgs.append(f"logits {err(Zrm, S * 1.4426950408889634):.2e} (max |S| {smax:.2e})")
        if err(lse * 0.6931471805599453, ref_lse) > 2e-6 * max(1.0, float(ref_lse.abs().max())):
            msgs.append(f"lse {err(lse * 0.6931471805599453, ref_lse):.2e}")
        for name, got, got32, ref in (("ce", ce, ce32, ref_ce), ("du_unit", du, du32, ref_du), ("dI", dI, dI32, ref_dI), ("dI recomputed", dIr, dI32, ref_dI)):
            e16, e32 = err(got, ref), err(got32, ref)
            if not e16 <= (16 if scale > 10 else 4) * e32 + 2e-6 * max(float(ref.abs().max()), 1e-30):
                msgs.append(f"{name}: {e16:.2e} vs the fp32 pair's {e32:.2e} (max |ref| {float(ref.abs().max()):.2e})")
        if msgs:
            bad += 1
            print("MISMATCH", what, "|", "; ".join(msgs), flush=True)
    except Exception as ex:  # a crash is a finding too
        bad += 1
        print("ERROR", what, "|", type(ex).__name__, str(ex)[:300], flush=True)
    n += 1
print(f"{n} cases, {bad} findings in {time.time() - t0:.0f} s")
