"""Time the in-batch-softmax CE kernels alone (MI355X).  Usage: python tools/bench_ce.py [M N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N
from two_tower_models_amd import ops

lib = N.load()
dev = torch.device("cuda:0")
shapes = [(8192, 8192), (8192, 65536)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
D = 128
for M, Nn in shapes:
    U = torch.randn(M, D, device=dev) * 0.3
    I = torch.randn(Nn, D, device=dev) * 0.3
    coef = torch.rand(M, device=dev) / M
    lse, ce = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dU, dI = torch.empty(M, D, device=dev), torch.empty(Nn, D, device=dev)
    wsn = lib.tt_inbatch_ce_workspace_bytes(M, Nn, D)
    ws = torch.empty(wsn, dtype=torch.uint8, device=dev)

    def fwd():
        N.check(lib.tt_inbatch_ce_fwd(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), ce.data_ptr(),
                                      ws.data_ptr(), wsn, N.stream()), "fwd")

    def bwd():
        N.check(lib.tt_inbatch_ce_bwd(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), coef.data_ptr(),
                                      dU.data_ptr(), D, dI.data_ptr(), D, ws.data_ptr(), wsn, N.stream()), "bwd")

    du_unit = torch.empty(M, D, device=dev)

    def fwd_du():  # forward + expected item embedding (the user-side gradient up to the row factor)
        N.check(lib.tt_inbatch_ce_fwd_du(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), ce.data_ptr(),
                                         du_unit.data_ptr(), D, ws.data_ptr(), wsn, N.stream()), "fwd_du")

    def bwd_items():  # item-side gradient only (dU == NULL)
        N.check(lib.tt_inbatch_ce_bwd(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), coef.data_ptr(),
                                      None, D, dI.data_ptr(), D, ws.data_ptr(), wsn, N.stream()), "bwd_items")

    zn = lib.tt_inbatch_ce_logits_bytes(M, Nn)
    Z = torch.empty(zn, dtype=torch.uint8, device=dev)
    dI2 = torch.empty(Nn, D, device=dev)

    def fwd_du_keep():  # the same, also writing the logits out
        N.check(lib.tt_inbatch_ce_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), ce.data_ptr(),
                                              du_unit.data_ptr(), D, Z.data_ptr(), zn, ws.data_ptr(), wsn, N.stream()),
                "fwd_du_keep")

    def bwd_kept():  # item-side gradient from the kept logits (no second U I^T product)
        N.check(lib.tt_inbatch_ce_bwd_kept(U.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), coef.data_ptr(), Z.data_ptr(), zn,
                                           dI2.data_ptr(), D, ws.data_ptr(), wsn, N.stream()), "bwd_kept")

    have16 = bool(lib.tt_ce16_supported(M, Nn, D))
    if have16:  # the split-fp16 pair (csrc/ce_f16x2.hip, exploratory)
        ws16n = lib.tt_ce16_workspace_bytes(M, Nn, D)
        ws16 = torch.empty(ws16n, dtype=torch.uint8, device=dev)
        Z16 = torch.empty(M * Nn, device=dev)
        dI16, du16, lse16, ce16 = torch.empty(Nn, D, device=dev), torch.empty(M, D, device=dev), torch.empty(M, device=dev), torch.empty(M, device=dev)

        def fwd_du_keep_f16x2():
            N.check(lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse16.data_ptr(), ce16.data_ptr(),
                                            du16.data_ptr(), D, Z16.data_ptr(), M * Nn * 4, ws16.data_ptr(), ws16n, N.stream()), "ce16 fwd")

        def bwd_kept_f16x2():
            N.check(lib.tt_ce16_bwd_kept(U.data_ptr(), D, M, Nn, D, 0, lse16.data_ptr(), coef.data_ptr(), Z16.data_ptr(), M * Nn * 4,
                                         dI16.data_ptr(), D, ws16.data_ptr(), ws16n, N.stream()), "ce16 bwd")

        dI16r = torch.empty(Nn, D, device=dev)

        def fwd_du_nokeep_f16x2():  # the same forward, no logits written
            N.check(lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse16.data_ptr(), ce16.data_ptr(),
                                            du16.data_ptr(), D, None, 0, ws16.data_ptr(), ws16n, N.stream()), "ce16 fwd")

        def bwd_recompute_f16x2():  # logits tiles formed again on the fp16 pipe (images of the forward reused)
            N.check(lib.tt_ce16_bwd_recompute(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse16.data_ptr(), coef.data_ptr(),
                                              dI16r.data_ptr(), D, ws16.data_ptr(), ws16n, 1, N.stream()), "ce16 bwd rc")

    fwd_du(); bwd_items(); fwd_du_keep(); bwd_kept()
    torch.cuda.synchronize()
    if have16:
        fwd_du_keep_f16x2(); bwd_kept_f16x2()
        fwd_du_nokeep_f16x2(); bwd_recompute_f16x2()
        torch.cuda.synchronize()
        print(f"M={M} N={Nn}: split-fp16 dI with recomputed logits vs with kept logits: {float((dI16r - dI16).abs().max() / dI16.abs().max()):.2e} (rel to max)", flush=True)
        print(f"M={M} N={Nn}: split-fp16 pair vs the fp32-MFMA pair: lse {float((lse16 - lse).abs().max()):.2e} abs, "
              f"du_unit {float((du16 - du_unit).abs().max() / du_unit.abs().max()):.2e}, "
              f"dI {float((dI16 - dI2).abs().max() / dI2.abs().max()):.2e} (rel to max)", flush=True)
    err = (dI2 - dI).abs().max().item() / max(dI.abs().max().item(), 1e-30)
    print(f"M={M} N={Nn}: kept-logits dI vs recomputed dI: max rel-to-max error {err:.2e}", flush=True)

    for name, fn, flops in ((("fwd", fwd, 2.0 * M * Nn * D), ("bwd", bwd, 8.0 * M * Nn * D),
                             ("fwd_du", fwd_du, 4.0 * M * Nn * D), ("bwd_items", bwd_items, 4.0 * M * Nn * D),
                             ("fwd_du_keep", fwd_du_keep, 4.0 * M * Nn * D), ("bwd_kept", bwd_kept, 2.0 * M * Nn * D))
                            + ((("fwd_du_keep_f16x2", fwd_du_keep_f16x2, 4.0 * M * Nn * D),
                                ("bwd_kept_f16x2", bwd_kept_f16x2, 2.0 * M * Nn * D),
                                ("fwd_du_nokeep_f16x2", fwd_du_nokeep_f16x2, 4.0 * M * Nn * D),
                                ("bwd_recompute_f16x2", bwd_recompute_f16x2, 4.0 * M * Nn * D)) if have16 else ())):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 5)
        ms = sorted(ts)[2]
        print(f"M={M} N={Nn} D={D} {name}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s "
              f"(dma={'off' if os.environ.get('TT_CE_NO_DMA') else 'on'})", flush=True)
