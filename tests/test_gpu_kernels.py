"""Kernel-level parity (run on MI355X via `pytest -m gpu`): every C-ABI entry point of
libtt_hotpath.so against the CPU oracle / a float64 restatement on seeded inputs,
including ragged shapes, strided views, duplicates and out-of-range ids."""
import math
import os

import numpy as np
import pytest
import torch

import fixture_gen as fg
from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def T():
    from two_tower_models_amd import _native as N
    from two_tower_models_amd import ops
    N.load()
    return ops, N


def g(shape, seed):
    return torch.from_numpy(fg.gaussianish(shape, seed))


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,Nn,K", [(32, 50, 20), (256, 128, 256), (300, 77, 41), (8192, 128, 8),
                                    (128, 384, 4100), (1000, 256, 130), (64, 64, 1)])
def test_gemm_layouts(T, layout, M, Nn, K):
    ops, N = T
    a_shape = (M, K) if layout != 2 else (K, M)
    b_shape = (Nn, K) if layout == 0 else (K, Nn)
    A, B = g(a_shape, 1), g(b_shape, 2)
    bias = g((Nn,), 3)
    ref = ((A if layout != 2 else A.t()).double() @ (B.t() if layout == 0 else B).double()) + bias.double()
    out = torch.empty(M, Nn, device=DEV)
    ops.gemm(layout, A.to(DEV), B.to(DEV), out, M, Nn, K, bias=bias.to(DEV))
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-6 * max(1.0, math.sqrt(K)) * float(ref.abs().max()), err


def test_gemm_epilogues_strided_and_accumulate(T):
    ops, N = T
    M, Nn, K = 200, 96, 72
    A, B = g((M, K + 8), 5)[:, 4:4 + K], g((Nn, K), 6)  # A is a column-slice view (ld > K, 16-B misaligned rows ok)
    ref = A.double() @ B.double().t()
    big = torch.zeros(M, 2 * Nn, device=DEV)
    out = big[:, Nn:]  # write into a column slice
    Ad = g((M, K + 8), 5).to(DEV)[:, 4:4 + K]
    ops.gemm(N.TT_GEMM_NT, Ad, B.to(DEV), out, M, Nn, K, epilogue=N.TT_EPI_RELU)
    assert torch.allclose(out.cpu().double(), ref.clamp(min=0), atol=1e-4)
    assert float(big[:, :Nn].abs().max()) == 0.0
    aux = g((M, Nn), 7)
    out2 = torch.empty(M, Nn, device=DEV)
    ops.gemm(N.TT_GEMM_NT, Ad, B.to(DEV), out2, M, Nn, K, epilogue=N.TT_EPI_RELU_MASK, aux=aux.to(DEV))
    assert torch.allclose(out2.cpu().double(), ref * (aux > 0), atol=1e-4)
    ops.gemm(N.TT_GEMM_NT, Ad, B.to(DEV), out2, M, Nn, K, accumulate=True)
    assert torch.allclose(out2.cpu().double(), ref * (aux > 0) + ref, atol=2e-4)


@pytest.mark.parametrize("rows,n_out,k_in", [(5000, 96, 40), (70000, 384, 128), (300, 130, 64), (33, 7, 5)])
def test_gemm_tn_with_fused_column_sum(T, rows, n_out, k_in):
    """tt_gemm_tn_colsum_f32: dW = dy^T x and db = column sums of dy from one pass (split-K and
    single-pass shapes, 64^2 and 128^2 tiles, ragged edges, strided operands)."""
    ops, N = T
    dy_full, x = g((rows, n_out + 4), 301), g((rows, k_in), 302)
    dy = dy_full[:, 4:]
    dW = torch.empty(n_out, k_in, device=DEV)
    dyd = dy_full.to(DEV)[:, 4:]
    _, db = ops.gemm_tn_colsum(dyd, x.to(DEV), dW)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    assert float((dW.cpu().double() - ref_w).abs().max()) < 3e-6 * math.sqrt(rows) * max(1.0, float(ref_w.abs().max()))
    assert float((db.cpu().double() - ref_b).abs().max()) < 3e-6 * math.sqrt(rows) * max(1.0, float(ref_b.abs().max()))
    # the fused product equals the plain TN GEMM bit for bit, and the sum is deterministic
    plain = torch.empty(n_out, k_in, device=DEV)
    ops.gemm(N.TT_GEMM_TN, dyd, x.to(DEV), plain, n_out, k_in, rows)
    assert torch.equal(plain, dW)
    _, db2 = ops.gemm_tn_colsum(dyd, x.to(DEV), torch.empty_like(dW))
    assert torch.equal(db, db2)


@pytest.mark.parametrize("rows,n_out", [(204800, 384), (50001, 384), (16400, 384), (8195, 384), (8192, 384),
                                        (50001, 256), (16400, 128)])
def test_gemm_tn_streaming_path(T, rows, n_out):
    """K >= 8192 rows, N = 128, M = 384 (M = 128 / 256 with TT_GEMM_TN_STREAM_ALL=1, else the tiled kernel):
    the register-resident streaming TN kernel (gemm_tn_stream.hip) -- ragged and odd row counts, strided
    operands, accumulate, determinism."""
    ops, N = T
    dy_full, x_full = g((rows, n_out + 8), 311), g((rows, 132), 312)
    dy, x = dy_full[:, 4:4 + n_out], x_full[:, :128]
    dyd, xd = dy_full.to(DEV)[:, 4:4 + n_out], x_full.to(DEV)[:, :128]
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    base = g((n_out, 128), 313)
    dW = base.to(DEV).clone()
    _, db = ops.gemm_tn_colsum(dyd, xd, dW, accumulate=True)
    tol = 3e-6 * math.sqrt(rows)
    assert float((dW.cpu().double() - ref_w - base.double()).abs().max()) < tol * max(1.0, float(ref_w.abs().max()))
    assert float((db.cpu().double() - ref_b).abs().max()) < tol * max(1.0, float(ref_b.abs().max()))
    dW2 = torch.empty(n_out, 128, device=DEV)
    _, db2 = ops.gemm_tn_colsum(dyd, xd, dW2)
    dW3 = torch.empty(n_out, 128, device=DEV)
    ops.gemm(N.TT_GEMM_TN, dyd, xd, dW3, n_out, 128, rows)
    assert torch.equal(dW2, dW3) and torch.equal(db, db2)
    assert float((dW2.cpu().double() - ref_w).abs().max()) < tol * max(1.0, float(ref_w.abs().max()))


def test_colsum(T):
    ops, N = T
    X = g((1000, 300), 9)
    got = ops.colsum(X.to(DEV)[:, 10:250])
    assert torch.allclose(got.cpu().double(), X[:, 10:250].double().sum(0), atol=1e-3)


# ------------------------------------------------------------------ gathers
@pytest.mark.parametrize("D", [128, 50, 2, 32])
def test_gather_rows_and_oob(T, D):
    ops, N = T
    table = g((777, D), 11)
    ids = torch.from_numpy(fg.uniform_ids((333,), 777, 12))
    out = torch.full((333, D + 4), -1.0, device=DEV)
    ops.gather_rows_into(table.to(DEV), ids.to(DEV), out[:, :D])
    assert torch.equal(out[:, :D].cpu(), table[ids])
    assert float(out[:, D:].min()) == -1.0
    bad = ids.clone()
    bad[5] = 777
    bad[9] = -1
    ops.gather_rows_into(table.to(DEV), bad.to(DEV), out[:, :D])
    assert float(out[5, :D].abs().max()) == 0.0
    with pytest.raises(IndexError):
        N.oob.poll(torch.device(DEV), blocking=True)
    N.oob.poll(torch.device(DEV), blocking=True)  # flag was cleared


# ------------------------------------------------------------------ in-batch softmax CE
@pytest.mark.parametrize("M,Nn,D,off", [(32, 32, 40, 0), (200, 200, 50, 0), (256, 256, 128, 0), (1000, 1000, 128, 0),
                                        (300, 300, 256, 0), (130, 700, 200, 400), (64, 64, 129, 0),
                                        (128, 512, 128, 256), (100, 300, 64, 200), (64, 64, 2, 0), (1, 1, 8, 0)])
def test_inbatch_ce_forward_backward(T, M, Nn, D, off):
    ops, N = T
    U = (g((M, D), 21) * 0.5).requires_grad_(True)
    I = (g((Nn, D), 22) * 0.5).requires_grad_(True)
    coef = g((M,), 23).abs() / M
    ce_ref = R.inbatch_rowwise_ce(U, I, off)
    (ce_ref * coef).sum().backward()
    Ud, Id = U.detach().to(DEV).requires_grad_(True), I.detach().to(DEV).requires_grad_(True)
    ce = ops.InBatchSoftmaxCE.apply(Ud, Id, off)
    assert torch.allclose(ce.cpu(), ce_ref.detach(), atol=2e-5, rtol=1e-5)
    (ce * coef.to(DEV)).sum().backward()
    scale = float(U.grad.abs().max())
    assert torch.allclose(Ud.grad.cpu(), U.grad, atol=1e-5 * scale + 1e-9, rtol=1e-4)
    assert torch.allclose(Id.grad.cpu(), I.grad, atol=1e-5 * float(I.grad.abs().max()) + 1e-9, rtol=1e-4)


def test_inbatch_ce_large_logits_stable(T):
    ops, N = T
    U, I = g((96, 128), 31) * 3.0, g((96, 128), 32) * 3.0  # logits ~ +-100
    ce = ops.InBatchSoftmaxCE.apply(U.to(DEV), I.to(DEV), 0)
    ref = R.inbatch_rowwise_ce(U.double(), I.double())
    assert torch.isfinite(ce).all()
    assert torch.allclose(ce.cpu().double(), ref, atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("M,Nn,D,off,scale", [(300, 300, 128, 0, 0.5), (130, 700, 128, 400, 0.5), (96, 96, 128, 0, 3.0),
                                               (77, 200, 40, 50, 1.0), (8192, 8192, 128, 0, 0.3)])
def test_inbatch_ce_fused_forward_gives_the_user_gradient(T, M, Nn, D, off, scale):
    """tt_inbatch_ce_fwd_du: row statistics identical in meaning to tt_inbatch_ce_fwd, and
    coef * du_unit == the dU of tt_inbatch_ce_bwd (whose dU kernel it replaces), including rows
    whose softmax is saturated (logits ~ +-100 at scale 3) and the split / ragged shapes;
    tt_inbatch_ce_bwd with dU == NULL still returns the same dI."""
    ops, N = T
    lib = N.load()
    U = (g((M, D), 61) * scale).to(DEV)
    I = (g((Nn, D), 62) * scale).to(DEV)
    coef = (g((M,), 63).abs() / M).to(DEV)
    wsp, wsn = ops._ws(torch.device(DEV), lib.tt_inbatch_ce_workspace_bytes(M, Nn, D), "ce_test")
    e = lambda *shape: torch.empty(*shape, device=DEV)
    lse1, ce1, lse2, ce2, du_unit = e(M), e(M), e(M), e(M), e(M, D)
    N.check(lib.tt_inbatch_ce_fwd(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse1.data_ptr(), ce1.data_ptr(),
                                  wsp, wsn, N.stream()), "fwd")
    dU, dI = e(M, D), e(Nn, D)
    N.check(lib.tt_inbatch_ce_bwd(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse1.data_ptr(), coef.data_ptr(),
                                  dU.data_ptr(), D, dI.data_ptr(), D, wsp, wsn, N.stream()), "bwd")
    N.check(lib.tt_inbatch_ce_fwd_du(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse2.data_ptr(), ce2.data_ptr(),
                                     du_unit.data_ptr(), D, wsp, wsn, N.stream()), "fwd_du")
    dI2 = e(Nn, D)
    N.check(lib.tt_inbatch_ce_bwd(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse2.data_ptr(), coef.data_ptr(),
                                  None, D, dI2.data_ptr(), D, wsp, wsn, N.stream()), "bwd items")
    assert torch.allclose(ce2, ce1, atol=2e-5 * max(1.0, scale * scale * 10), rtol=1e-5)
    dU2 = du_unit * coef.unsqueeze(1)
    # fp64 reference for the user-side gradient
    Ud, Id = U.cpu().double(), I.cpu().double()
    P = torch.softmax(Ud @ Id.t(), dim=1)
    P[torch.arange(M), torch.arange(M) + off] -= 1.0
    ref = (P * coef.cpu().double().unsqueeze(1)) @ Id
    tol = 2e-5 * float(ref.abs().max()) + 1e-12
    assert float((dU2.cpu().double() - ref).abs().max()) <= tol
    assert float((dU.cpu().double() - ref).abs().max()) <= tol
    assert torch.allclose(dI2, dI, atol=1e-6 * float(dI.abs().max()) + 1e-12, rtol=1e-5)


@pytest.mark.parametrize("M,Nn,D,off,scale", [(300, 300, 128, 0, 0.5), (130, 700, 128, 400, 0.5), (96, 96, 128, 0, 3.0),
                                               (77, 200, 64, 50, 1.0), (1, 1, 32, 0, 1.0), (129, 1000, 32, 871, 1.0),
                                               (2048, 16384, 128, 4096, 0.3)])
def test_inbatch_ce_kept_logits_backward(T, M, Nn, D, off, scale):
    """tt_inbatch_ce_fwd_du_keep / tt_inbatch_ce_bwd_kept (logits written out by the forward, item-side
    gradient rebuilt from them): same row statistics and du_unit as tt_inbatch_ce_fwd_du, same dI as the
    recomputing backward and as the oracle -- ragged M / N (padded buffer), non-zero diagonal offset,
    split streams, saturated softmax rows."""
    ops, N = T
    lib = N.load()
    if os.environ.get("TT_CE_NO_DMA"):
        pytest.skip("the kept-logits kernels exist in the LDS-DMA form only (A/B switch TT_CE_NO_DMA is set)")
    U = (g((M, D), 81) * scale).to(DEV)
    I = (g((Nn, D), 82) * scale).to(DEV)
    coef = (g((M,), 83).abs() / M).to(DEV)
    wsp, wsn = ops._ws(torch.device(DEV), lib.tt_inbatch_ce_workspace_bytes(M, Nn, D), "ce_test")
    e = lambda *shape: torch.empty(*shape, device=DEV)
    lse1, ce1, du1, lse2, ce2, du2 = e(M), e(M), e(M, D), e(M), e(M), e(M, D)
    N.check(lib.tt_inbatch_ce_fwd_du(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse1.data_ptr(), ce1.data_ptr(),
                                     du1.data_ptr(), D, wsp, wsn, N.stream()), "fwd_du")
    dI1 = e(Nn, D)
    N.check(lib.tt_inbatch_ce_bwd(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse1.data_ptr(), coef.data_ptr(),
                                  None, D, dI1.data_ptr(), D, wsp, wsn, N.stream()), "bwd items")
    zn = lib.tt_inbatch_ce_logits_bytes(M, Nn)
    assert zn >= M * Nn * 4
    Z = torch.full((zn // 4,), float("nan"), device=DEV)  # whatever the kernels do not write must not matter
    N.check(lib.tt_inbatch_ce_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse2.data_ptr(), ce2.data_ptr(),
                                          du2.data_ptr(), D, Z.data_ptr(), zn, wsp, wsn, N.stream()), "fwd_du_keep")
    dI2 = e(Nn, D)
    N.check(lib.tt_inbatch_ce_bwd_kept(U.data_ptr(), D, M, Nn, D, off, lse2.data_ptr(), coef.data_ptr(), Z.data_ptr(), zn,
                                       dI2.data_ptr(), D, wsp, wsn, N.stream()), "bwd_kept")
    assert torch.equal(ce2, ce1) and torch.equal(lse2, lse1) and torch.equal(du2, du1)
    assert torch.isfinite(dI2).all()
    assert torch.allclose(dI2, dI1, atol=1e-6 * float(dI1.abs().max()) + 1e-12, rtol=1e-5)
    # the kept logits themselves: log2-domain U I^T in the top-left M x N block
    ldk = (Nn + 127) // 128 * 128
    S = (U.cpu().double() @ I.cpu().double().t()) * 1.4426950408889634
    got = Z.view(-1, ldk)[:M, :Nn].cpu().double()
    assert float((got - S).abs().max()) <= 1e-5 * max(1.0, float(S.abs().max()))
    # and the oracle's gradient
    P = torch.softmax(U.cpu().double() @ I.cpu().double().t(), dim=1)
    P[torch.arange(M), torch.arange(M) + off] -= 1.0
    ref = (P * coef.cpu().double().unsqueeze(1)).t() @ U.cpu().double()
    assert float((dI2.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-12
    # too small a buffer / unsupported width are refused, not overrun
    assert lib.tt_inbatch_ce_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse2.data_ptr(), ce2.data_ptr(),
                                         du2.data_ptr(), D, Z.data_ptr(), zn - 4, wsp, wsn, N.stream()) != 0


@pytest.mark.parametrize("M,Nn,off,scale", [(256, 1024, 0, 0.5), (256, 2048, 1536, 0.35), (512, 1024, 512, 3.0),
                                             (1024, 8192, 4096, 0.3), (2048, 2048, 0, 0.4)])
def test_split_fp16_ce_pair_vs_float64(T, M, Nn, off, scale):
    """tt_ce16_fwd_du_keep / tt_ce16_bwd_kept (csrc/ce_f16x2.hip, exploratory and opt-in): the kept-logits pair with every
    product as three fp16 MFMA products of two-term splits.  Against float64: logits, lse, ce, unit user gradient, item
    gradient -- each also next to the fp32-MFMA pair's error on the same inputs (the split form must stay within 4x of it
    plus a floor: 'fp32-grade'), non-zero diagonal offset, saturated softmax rows (scale 3), and the shapes it refuses."""
    ops, N = T
    lib = N.load()
    D = 128
    assert lib.tt_ce16_supported(M, Nn, D) == 1
    assert lib.tt_ce16_supported(M, Nn + 8, D) == 0 and lib.tt_ce16_supported(M + 1, Nn, D) == 0 and lib.tt_ce16_supported(M, Nn, 64) == 0
    U = (g((M, D), 181) * scale).to(DEV)
    I = (g((Nn, D), 182) * scale).to(DEV)
    coef = (g((M,), 183) * (37.0 if off == 512 else 1.0 / M)).to(DEV)  # any sign and size (scaled by its own maximum inside)
    e = lambda *shape: torch.full(shape, float("nan"), device=DEV)
    wsn = lib.tt_ce16_workspace_bytes(M, Nn, D)
    ws = torch.empty(wsn, dtype=torch.uint8, device=DEV)
    lse, ce, du, Z, dI = e(M), e(M), e(M, D), e(M, Nn), e(Nn, D)
    for _ in range(2):  # twice: nothing in the workspace may carry over
        N.check(lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), ce.data_ptr(), du.data_ptr(), D,
                                        Z.data_ptr(), M * Nn * 4, ws.data_ptr(), wsn, N.stream()), "ce16 fwd")
        N.check(lib.tt_ce16_bwd_kept(U.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), coef.data_ptr(), Z.data_ptr(), M * Nn * 4,
                                     dI.data_ptr(), D, ws.data_ptr(), wsn, N.stream()), "ce16 bwd")
    # the recomputing form: no logits buffer at all; once trusting the forward's images in the workspace, once forming them again
    lse_r, ce_r, du_r, dI_r, dI_r2 = e(M), e(M), e(M, D), e(Nn, D), e(Nn, D)
    N.check(lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse_r.data_ptr(), ce_r.data_ptr(), du_r.data_ptr(), D,
                                    None, 0, ws.data_ptr(), wsn, N.stream()), "ce16 fwd (no logits)")
    N.check(lib.tt_ce16_bwd_recompute(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse_r.data_ptr(), coef.data_ptr(), dI_r.data_ptr(), D,
                                      ws.data_ptr(), wsn, 1, N.stream()), "ce16 bwd recompute (reuse)")
    ws.fill_(0xA5)
    N.check(lib.tt_ce16_bwd_recompute(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse_r.data_ptr(), coef.data_ptr(), dI_r2.data_ptr(), D,
                                      ws.data_ptr(), wsn, 0, N.stream()), "ce16 bwd recompute")
    assert torch.equal(lse_r, lse) and torch.equal(ce_r, ce) and torch.equal(du_r, du)  # the forward's results do not depend on keeping
    assert torch.equal(dI_r, dI_r2)
    # the fp32-MFMA pair on the same inputs
    wsp, wsn32 = ops._ws(torch.device(DEV), lib.tt_inbatch_ce_workspace_bytes(M, Nn, D), "ce_test")
    lse32, ce32, du32, dI32 = e(M), e(M), e(M, D), e(Nn, D)
    zn = lib.tt_inbatch_ce_logits_bytes(M, Nn)
    Z32 = torch.empty(zn // 4, device=DEV)
    N.check(lib.tt_inbatch_ce_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse32.data_ptr(), ce32.data_ptr(),
                                          du32.data_ptr(), D, Z32.data_ptr(), zn, wsp, wsn32, N.stream()), "fwd_du_keep")
    N.check(lib.tt_inbatch_ce_bwd_kept(U.data_ptr(), D, M, Nn, D, off, lse32.data_ptr(), coef.data_ptr(), Z32.data_ptr(), zn,
                                       dI32.data_ptr(), D, wsp, wsn32, N.stream()), "bwd_kept")
    Ud, Id, cd = U.cpu().double(), I.cpu().double(), coef.cpu().double()
    S = Ud @ Id.t()
    rows = torch.arange(M)
    ref_lse = torch.logsumexp(S, dim=1)
    ref_ce = ref_lse - S[rows, rows + off]
    P = torch.softmax(S, dim=1)
    ref_du = P @ Id - Id[rows + off]
    P[rows, rows + off] -= 1.0
    ref_dI = (P * cd.unsqueeze(1)).t() @ Ud

    def err(got, ref):
        return float((got.cpu().double() - ref).abs().max())

    smax = max(1.0, float(S.abs().max()))
    Zrm = Z.view(M // 32, Nn // 32, 32, 32).permute(0, 2, 1, 3).reshape(M, Nn)  # tiles of 32 x 32, row-major inside
    assert err(Zrm, S * 1.4426950408889634) <= 2e-6 * smax  # log2-domain logits, every element
    assert err(lse * 0.6931471805599453, ref_lse) <= 2e-6 * max(1.0, float(ref_lse.abs().max()))  # row_lse: log2 domain, like the fp32 pair's
    for name, got, got32, ref in (("ce", ce, ce32, ref_ce), ("du_unit", du, du32, ref_du), ("dI", dI, dI32, ref_dI),
                                  ("dI (recomputed logits)", dI_r, dI32, ref_dI)):
        e16, e32 = err(got, ref), err(got32, ref)
        floor = 2e-6 * max(float(ref.abs().max()), 1e-30)
        assert e16 <= 4 * e32 + floor, (name, e16, e32, float(ref.abs().max()))
    loss = float((ref_ce * cd).sum())
    assert abs(float((ce.double().cpu() * cd).sum()) - loss) < 1e-6 * max(1.0, abs(loss))  # the loss: far inside 1e-4
    # refused, not overrun
    assert lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), ce.data_ptr(), du.data_ptr(), D,
                                   Z.data_ptr(), M * Nn * 4 - 4, ws.data_ptr(), wsn, N.stream()) != 0
    assert lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), ce.data_ptr(), du.data_ptr(), D,
                                   Z.data_ptr(), M * Nn * 4, ws.data_ptr(), wsn - 256, N.stream()) == N.TT_E_WORKSPACE
    assert lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn - 8, D, 0, lse.data_ptr(), ce.data_ptr(), du.data_ptr(), D,
                                   Z.data_ptr(), M * Nn * 4, ws.data_ptr(), wsn, N.stream()) == N.TT_E_UNSUPPORTED


def test_inbatch_ce_op_keep_logits_matches_default(T):
    """ops.InBatchSoftmaxCE(keep_logits=True) == the default path through autograd, and the default
    switches itself on for wide negative sets only."""
    ops, N = T
    M, Nn, D, off = 256, 1024, 128, 512
    U0, I0 = g((M, D), 91) * 0.5, g((Nn, D), 92) * 0.5
    coef = (g((M,), 93).abs() / M).to(DEV)
    grads = []
    for keep in (False, True, None):
        U, I = U0.to(DEV).requires_grad_(True), I0.to(DEV).requires_grad_(True)
        ce = ops.InBatchSoftmaxCE.apply(U, I, off, keep)
        (ce * coef).sum().backward()
        grads.append((ce.detach(), U.grad, I.grad))
    for k in (1, 2):
        assert torch.equal(grads[k][0], grads[0][0]) and torch.equal(grads[k][1], grads[0][1])
        assert torch.allclose(grads[k][2], grads[0][2], atol=1e-6 * float(grads[0][2].abs().max()), rtol=1e-5)
    # D = 40 has no kept-logits kernel: the request falls back to the recomputing pair
    U, I = (g((64, 40), 94)).to(DEV).requires_grad_(True), (g((64, 40), 95)).to(DEV).requires_grad_(True)
    ops.InBatchSoftmaxCE.apply(U, I, 0, True).sum().backward()
    assert torch.isfinite(I.grad).all()


def test_dx_product_with_pool_backward_epilogue(T):
    """tt_hist_dx_pool_bwd: dx = dQKV W_in + d_pooled / H in one pass (the POOL form of gemm_ws16_kernel) against float64,
    strided d_pooled rows (slot 1 of the [B, 2, D] cotangent), a ragged last stage; small problems are refused."""
    ops, N = T
    lib = N.load()
    B, H, D = 700, 29, 128  # B * H = 20 300 rows: not a multiple of the 32-row stage
    dqkv, w_in = g((B * H, 3 * D), 411) * 0.3, g((3 * D, D), 412) / math.sqrt(D)
    cot = g((B, 2 * D), 413)
    dq, w, c = dqkv.to(DEV), w_in.to(DEV), cot.to(DEV)
    dx = torch.full((B * H, D), float("nan"), device=DEV)
    N.check(lib.tt_hist_dx_pool_bwd(dq.data_ptr(), w.data_ptr(), B, H, D, c[:, D:].data_ptr(), 2 * D, dx.data_ptr(), N.stream()), "dx_pool")
    want = dqkv.double() @ w_in.double() + (cot[:, D:].double() / H).repeat_interleave(H, dim=0)
    assert float((dx.cpu().double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    assert lib.tt_hist_dx_pool_bwd(dq.data_ptr(), w.data_ptr(), 4, H, D, c[:, D:].data_ptr(), 2 * D, dx.data_ptr(), N.stream()) == N.TT_E_UNSUPPORTED


def test_inbatch_ce_op_through_split_fp16_pair(T, monkeypatch):
    """ops.InBatchSoftmaxCE with TT_CE_F16X2 (EXPLORATORY): the autograd op takes the split-fp16 pair where its shapes allow
    and lands on the fp32-MFMA path's values to within the pair's error (test_split_fp16_ce_pair_vs_float64); unsupported
    shapes and no-grad calls keep the fp32 path bit for bit."""
    ops, N = T
    M, Nn, D, off = 512, 2048, 128, 1024
    U0, I0 = g((M, D), 191) * 0.5, g((Nn, D), 192) * 0.5
    coef = (g((M,), 193).abs() / M).to(DEV)
    res = []
    for split in (False, True):
        monkeypatch.setattr(ops, "_CE_F16X2", split)
        assert ops.ce16_usable(U0.to(DEV), I0.to(DEV)) == split
        U, I = U0.to(DEV).requires_grad_(True), I0.to(DEV).requires_grad_(True)
        ce = ops.InBatchSoftmaxCE.apply(U, I, off)
        (ce * coef).sum().backward()
        res.append((ce.detach(), U.grad, I.grad))
    assert not torch.equal(res[0][0], res[1][0])  # a different kernel did run
    for a, b in zip(res[0], res[1]):
        assert torch.allclose(a, b, atol=3e-6 * float(a.abs().max()), rtol=1e-5)
    # shapes the pair does not take (M = 300) and inference calls: the fp32 path, untouched by the switch
    monkeypatch.setattr(ops, "_CE_F16X2", True)
    U, I = (g((300, D), 194)).to(DEV), (g((2048, D), 195)).to(DEV)
    assert not ops.ce16_usable(U, I) and ops.ce16_usable(I[:1024], I[:1024])
    with torch.no_grad():
        ce_a = ops.InBatchSoftmaxCE.apply(U0.to(DEV), I0.to(DEV), off)
    monkeypatch.setattr(ops, "_CE_F16X2", False)
    with torch.no_grad():
        ce_b = ops.InBatchSoftmaxCE.apply(U0.to(DEV), I0.to(DEV), off)
    assert torch.equal(ce_a, ce_b)


@pytest.mark.parametrize("B,Tn,DI,case", [(64, 1, 128, "plain"), (1000, 3, 40, "plain"), (8192, 1, 128, "plain"),
                                          (300, 1, 64, "all_zero_labels"), (257, 2, 32, "clamped_priors"), (1, 1, 8, "plain")])
def test_debias_loss_head_matches_torch_expressions(T, B, Tn, DI, case):
    """ops.DebiasedWeightedLoss (tt_debias_loss_fwd / _bwd) against the reference's tensor expressions
    (ref:src/two_tower_with_debiasing.py:77-129 + ref:src/two_tower_base_retrieval.py:322-345) evaluated
    in float64 on the CPU: loss and every gradient, including ties at the batch maximum (0/1 labels make
    many equal weights impossible only after the division -- the all-zero case ties EVERY row), priors
    below the 1e-3 clamp, and the [B,1]-vs-[B] broadcast of the position term."""
    import warnings
    ops, N = T
    row_ce = (g((B,), 101).abs() * 3 + 0.1)
    labels = (fg.hashed_u64((B, Tn), 102) % np.uint64(2)).astype(np.float32)
    if case == "all_zero_labels":
        labels[:] = 0.0
    labels = torch.from_numpy(labels)
    uvw = torch.tensor([1.0, 0.5, 2.0][:Tn])
    position = torch.from_numpy((fg.hashed_u64((B,), 103) % np.uint64(100)).astype(np.int64))
    ue = g((B, DI), 104) * 0.5
    pos_table = g((100, 1), 105) * 0.3 + (0.0 if case == "clamped_priors" else 0.8)
    lin_w = g((1, DI + 1), 106) * (0.02 if case != "clamped_priors" else 0.2)
    lin_b = torch.tensor([0.7 if case != "clamped_priors" else -0.2])

    def reference(row_ce, ue, pos_table, lin_w, lin_b):
        nuv = torch.sum(labels.double() * uvw.double(), dim=-1)
        p = pos_table[position]                                            # [B, 1]
        e = (torch.cat([ue, p], dim=-1) @ lin_w.t() + lin_b).squeeze(1)    # [B]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            aux = torch.nn.functional.mse_loss(e, nuv, reduction="sum") + torch.nn.functional.mse_loss(p, nuv, reduction="sum")
        w = nuv / torch.clamp(e, min=1e-3)
        w = torch.clamp(w, min=0.000001)
        w = w / torch.max(w)
        return torch.mean(row_ce * w) + aux

    leaves64 = [t.double().requires_grad_(True) for t in (row_ce, ue, pos_table, lin_w, lin_b)]
    want = reference(*leaves64)
    want.backward()
    leaves = [t.to(DEV).requires_grad_(True) for t in (row_ce, ue, pos_table, lin_w, lin_b)]
    got = ops.DebiasedWeightedLoss.apply(leaves[0], labels.to(DEV), uvw.to(DEV), position.to(DEV), *leaves[1:])
    assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    got.backward()
    for name, a, b in zip(("row_ce", "user_emb", "pos_table", "lin_w", "lin_b"), leaves, leaves64):
        scale = float(b.grad.abs().max())
        assert torch.allclose(a.grad.cpu().double(), b.grad, atol=2e-5 * scale + 1e-12, rtol=2e-4), (name, case)
    # bit-reproducible: fixed-order reductions
    leaves2 = [t.to(DEV).requires_grad_(True) for t in (row_ce, ue, pos_table, lin_w, lin_b)]
    got2 = ops.DebiasedWeightedLoss.apply(leaves2[0], labels.to(DEV), uvw.to(DEV), position.to(DEV), *leaves2[1:])
    got2.backward()
    assert torch.equal(got2, got) and all(torch.equal(a.grad, b.grad) for a, b in zip(leaves, leaves2))
    # an out-of-range position is reported like every other out-of-range id
    bad = position.clone()
    bad[0] = 100
    ops.DebiasedWeightedLoss.apply(leaves[0].detach(), labels.to(DEV), uvw.to(DEV), bad.to(DEV), *[t.detach() for t in leaves[1:]])
    with pytest.raises(IndexError):
        N.oob.poll(torch.device(DEV), blocking=True)


@pytest.mark.parametrize("mode", ["position", "user"])
@pytest.mark.parametrize("B,Tn,DI,case", [(64, 1, 128, "plain"), (1000, 3, 40, "plain"), (8192, 1, 128, "plain"),
                                          (300, 1, 64, "all_zero_labels"), (257, 2, 32, "clamped_priors"), (1, 1, 8, "plain")])
def test_single_term_debias_heads_match_torch_expressions(T, B, Tn, DI, case, mode):
    """The fused head's position-only / user-only modes (tt_debias_loss_fwd(mode = TT_DEBIAS_POSITION | TT_DEBIAS_USER))
    against the two sibling hooks' tensor expressions (ref:src/two_tower_with_position_debiased_weights.py:95-113: the MSE
    sees the RAW prior, the division its 1e-3-clamped copy; ref:src/two_tower_with_user_debiased_weights.py:121-135: the
    1e-1 clamp comes FIRST, a clamped row sends the head no gradient) on ref:src/two_tower_base_retrieval.py:334-343, in
    float64: loss and every gradient, incl. all-tied weights and priors below their clamp."""
    ops, N = T
    row_ce = (g((B,), 111).abs() * 3 + 0.1)
    labels = (fg.hashed_u64((B, Tn), 112) % np.uint64(2)).astype(np.float32)
    if case == "all_zero_labels":
        labels[:] = 0.0
    labels = torch.from_numpy(labels)
    uvw = torch.tensor([1.0, 0.5, 2.0][:Tn])
    position = torch.from_numpy((fg.hashed_u64((B,), 113) % np.uint64(100)).astype(np.int64))
    ue = g((B, DI), 114) * 0.5
    pos_table = g((100, 1), 115) * 0.3 + (0.0 if case == "clamped_priors" else 0.8)
    lin_w = g((1, DI), 116) * (0.02 if case != "clamped_priors" else 0.2)
    lin_b = torch.tensor([0.7 if case != "clamped_priors" else 0.05])

    def reference(row_ce, ue, pos_table, lin_w, lin_b):
        nuv = torch.sum(labels.double() * uvw.double(), dim=-1)
        if mode == "position":
            prior = pos_table[position, 0]
            aux = torch.sum((prior - nuv) ** 2)
            w = nuv / prior.clamp(min=1e-3)
        else:
            prior = (ue @ lin_w[0] + lin_b[0]).clamp(min=1e-1)
            aux = torch.sum((prior - nuv) ** 2)
            w = nuv / prior
        w = torch.clamp(w, min=0.000001)
        w = w / torch.max(w)
        return torch.mean(row_ce * w) + aux

    leaves64 = [t.double().requires_grad_(True) for t in (row_ce, ue, pos_table, lin_w, lin_b)]
    want = reference(*leaves64)
    want.backward()
    leaves = [t.to(DEV).requires_grad_(True) for t in (row_ce, ue, pos_table, lin_w, lin_b)]

    def run(lv):
        if mode == "position":
            return ops.DebiasedWeightedLoss.apply(lv[0], labels.to(DEV), uvw.to(DEV), position.to(DEV), lv[1], lv[2], None, None,
                                                  N.TT_DEBIAS_POSITION)
        return ops.DebiasedWeightedLoss.apply(lv[0], labels.to(DEV), uvw.to(DEV), position.to(DEV), lv[1], None, lv[3], lv[4],
                                              N.TT_DEBIAS_USER)

    got = run(leaves)
    assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    got.backward()
    used = {"position": ("row_ce", "user_emb", "pos_table"), "user": ("row_ce", "user_emb", "lin_w", "lin_b")}[mode]
    for name, a, b in zip(("row_ce", "user_emb", "pos_table", "lin_w", "lin_b"), leaves, leaves64):
        if name not in used:
            assert a.grad is None
            continue
        want_g = b.grad if b.grad is not None else torch.zeros_like(b)
        scale = float(want_g.abs().max())
        assert torch.allclose(a.grad.cpu().double(), want_g, atol=2e-5 * scale + 1e-12, rtol=2e-4), (name, case, mode)
    leaves2 = [t.to(DEV).requires_grad_(True) for t in (row_ce, ue, pos_table, lin_w, lin_b)]
    got2 = run(leaves2)
    got2.backward()
    assert torch.equal(got2, got) and all(a.grad is None or torch.equal(a.grad, b.grad) for a, b in zip(leaves, leaves2))


def test_weighted_mean_loss(T):
    ops, N = T
    B, Tn = 777, 3
    ce = g((B,), 41).abs()
    labels = (torch.from_numpy(fg.hashed_u64((B, Tn), 42) % np.uint64(2)).float())
    uvw = torch.tensor([0.1, 0.2, 0.3])
    ced = ce.to(DEV).requires_grad_(True)
    loss = ops.WeightedMeanLoss.apply(ced, labels.to(DEV), uvw.to(DEV))
    w = R.normalise_value_weights(R.net_user_value(labels, uvw))
    assert abs(loss.item() - float((ce * w).mean())) < 1e-6
    loss.backward()
    assert torch.allclose(ced.grad.cpu(), w / B, atol=1e-8)


@pytest.mark.parametrize("M,D,Tn", [(8192, 128, 3), (4096, 128, 0), (777, 64, 3), (17, 32, 1), (300, 256, 3)])
def test_inbatch_ce_with_fused_loss_head_is_the_two_op_path_bit_for_bit(T, M, D, Tn):
    """tt_inbatch_ce_fwd_du_loss + tt_scale_rows_g (loss head inside the forward's finishing launch, run by whichever
    workgroup arrives last) against InBatchSoftmaxCE + WeightedMeanLoss: the loss and every gradient are IDENTICAL, run
    after run (the head reads complete arrays in a fixed order); ragged row counts (not a multiple of 16), labels None
    (train.py's 1-D labels), the D > 128 generic form; and the loss against the CPU oracle within 1e-5."""
    ops, N = T
    gen = torch.Generator().manual_seed(M + D)
    U0, I0 = torch.randn(M, D, generator=gen) * 0.4, torch.randn(M, D, generator=gen) * 0.4
    labels = (torch.rand(M, Tn, generator=gen) < 0.4).float() if Tn else None
    uvw = torch.tensor([0.1, 0.2, 0.3][:max(Tn, 1)])
    labd, uvwd = (labels.to(DEV) if Tn else None), uvw.to(DEV)
    outs = []
    for fused in (True, False, True):
        U, I = U0.clone().to(DEV).requires_grad_(True), I0.clone().to(DEV).requires_grad_(True)
        if fused:
            assert ops.fused_loss_supported(U, I, labd, uvwd)
            loss = ops.InBatchSoftmaxWeightedLoss.apply(U, I, labd, uvwd)
        else:
            loss = ops.WeightedMeanLoss.apply(ops.InBatchSoftmaxCE.apply(U, I, 0), labd, uvwd)
        (loss * 1.75).backward()  # an upstream gradient other than 1
        outs.append((loss.detach().clone(), U.grad.clone(), I.grad.clone()))
    for a, b in ((outs[0], outs[1]), (outs[0], outs[2])):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    ce = R.inbatch_rowwise_ce(U0, I0)
    w = R.normalise_value_weights(R.net_user_value(labels, uvw)) if Tn else torch.ones(M)
    assert abs(outs[0][0].item() - float((ce * w).mean())) < 1e-5 * max(1.0, float(ce.mean()))


# ------------------------------------------------------------------ row plan + Adam
@pytest.mark.parametrize("n,n_rows", [(1, 10), (777, 100), (8192, 1_000_000), (50_000, 300), (4096, 70_000_000),
                                      (63, 2), (1025, 5000), (8191, 3), (8193, 1_000_000), (5000, 1)])
def test_rowgrad_plan_matches_stable_sort(T, n, n_rows):
    ops, N = T
    ids = torch.from_numpy(fg.uniform_ids((n,), n_rows, 51))
    rows = torch.zeros(n, 4, device=DEV)
    plan = ops.RowPlan.from_grads([ops.RowGrad(ids.to(DEV), rows)], n_rows)
    order = torch.sort(ids, stable=True)
    assert torch.equal(plan.sorted_ids.cpu().long(), order.values)
    assert torch.equal(plan.perm.cpu().long(), order.indices)
    uniq, counts = torch.unique_consecutive(order.values, return_counts=True)
    U = int(plan.n_unique.item())
    assert U == len(uniq)
    begins = torch.cumsum(counts, 0) - counts
    assert torch.equal(plan.seg_begin.cpu()[:U].long(), begins)
    assert int(plan.seg_begin[U]) == n


@pytest.mark.parametrize("n_rows,D,n1,n2", [(300, 128, 256, 512), (97, 50, 64, 0), (5000, 32, 1000, 3000)])
def test_adam_table_equals_dense_adam(T, n_rows, D, n1, n2):
    """3 dense-exact steps on a table looked up through two id blocks (item ids + history
    ids) == the oracle's Adam on the dense gradient, for touched AND untouched rows."""
    import ctypes as C
    ops, N = T
    lib = N.load()
    W = g((n_rows, D), 61)
    p_ref, m_ref, v_ref = W.clone(), torch.zeros_like(W), torch.zeros_like(W)
    Wd, Md, Vd = W.to(DEV), torch.zeros(n_rows, D, device=DEV), torch.zeros(n_rows, D, device=DEV)
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0, 0, 0, 0], dtype=torch.float64, device=DEV)
    for step in range(1, 4):
        blocks, dense = [], torch.zeros(n_rows, D)
        for k, n in enumerate((n1, n2)):
            if n == 0:
                continue
            ids = torch.from_numpy(fg.uniform_ids((n,), n_rows, 70 + 10 * step + k))
            rows = g((n, D), 80 + 10 * step + k) * 0.01
            dense.index_add_(0, ids, rows)
            blocks.append(ops.RowGrad(ids.to(DEV), rows.to(DEV)))
        R.adam_update(p_ref, dense, m_ref, v_ref, step)
        N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "advance")
        plan = ops.RowPlan.from_grads(blocks, n_rows)
        wsp, wsn = ops._ws(torch.device(DEV), lib.tt_adam_table_workspace_bytes(plan.n, D), "adam_side")
        N.check(lib.tt_adam_table(Wd.data_ptr(), Md.data_ptr(), Vd.data_ptr(), n_rows, D, hyper.data_ptr(),
                                  C.byref(plan.sources), plan.n, plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                  plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), wsp, wsn, N.stream()), "adam")
        dg = ops.dense_grad_from_rows(blocks, n_rows, D)
        assert torch.allclose(dg.cpu(), dense, atol=1e-6)
    assert torch.allclose(Wd.cpu(), p_ref, atol=3e-6, rtol=1e-5)
    assert torch.allclose(Md.cpu(), m_ref, atol=1e-7, rtol=1e-4)
    assert torch.allclose(Vd.cpu(), v_ref, atol=1e-9, rtol=1e-4)
    assert float(hyper[4]) == 3.0


@pytest.mark.parametrize("n_rows,D,n", [(5003, 128, 900), (1000, 64, 700), (777, 32, 300), (300, 256, 100), (40000, 128, 30000)])
def test_adam_marked_sweep_equals_the_single_call(T, n_rows, D, n):
    """The marked schedule (tt_adam_mark_rows, tt_adam_tables_sweep_marked, finish with side == NULL: the sweep steps over
    the looked-up rows and the finish reads their old p, m, v from the table) vs tt_adam_table (park, sweep, write back) on
    the same tables: p, m, v bit for bit -- duplicate ids, ids outside the block, row counts that are no multiple of 32 or
    of a chunk, a second table of the same launch with no marks at all."""
    import ctypes as C
    ops, N = T
    lib = N.load()
    assert lib.tt_adam_marked_supported(D) == 1 and lib.tt_adam_marked_supported(96) == 0 and lib.tt_adam_marked_supported(16) == 0
    g2 = torch.Generator().manual_seed(n_rows + D)
    W0 = torch.randn(n_rows, D, generator=g2)
    M0, V0 = torch.randn(n_rows, D, generator=g2) * 0.01, torch.rand(n_rows, D, generator=g2) * 1e-4
    X0 = torch.randn(3 * 1024 + 40, generator=g2)  # the unmarked second table: three chunks and a bit
    ids = torch.randint(0, n_rows, (n,), generator=g2)
    ids[:7] = ids[7:14]
    ids[20] = n_rows + 5
    ids[21] = n_rows - 1
    ids[22] = 0
    rows = (torch.randn(n, D, generator=g2) * 0.01).to(DEV)
    ids = ids.to(DEV)

    def fresh():
        hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 6, 0, 0, 0], dtype=torch.float64, device=DEV)
        N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "advance")
        plan = ops.RowPlan([ids], n_rows + 8)
        plan.attach([rows])
        return hyper, [t.clone().to(DEV) for t in (W0, M0, V0)], [X0.clone().to(DEV) for _ in range(3)], plan

    hyper, (W, M, V), (X, Y, Z), plan = fresh()
    wsn = lib.tt_adam_table_workspace_bytes(n, D)
    ws = torch.empty(wsn, dtype=torch.uint8, device=DEV)
    N.check(lib.tt_adam_table(W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, D, hyper.data_ptr(), C.byref(plan.sources), n,
                              plan.sorted_ids.data_ptr(), plan.perm.data_ptr(), plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(),
                              ws.data_ptr(), wsn, N.stream()), "tt_adam_table")
    N.check(lib.tt_adam_table_sweep(X.data_ptr(), Y.data_ptr(), Z.data_ptr(), X.numel() // 8, 8, hyper.data_ptr(), N.stream()), "sweep")

    hyper2, (W2, M2, V2), (X2, Y2, Z2), plan2 = fresh()
    words = lib.tt_adam_marks_words(n_rows)
    marks = torch.full((words,), -1, dtype=torch.int32, device=DEV)  # stale bits everywhere: mark_rows clears first
    N.check(lib.tt_adam_mark_rows(ids.data_ptr(), n, n_rows, marks.data_ptr(), words, N.stream()), "mark_rows")
    descs = (N.AdamTensor * 2)()
    descs[0].p, descs[0].m, descs[0].v, descs[0].n = W2.data_ptr(), M2.data_ptr(), V2.data_ptr(), W2.numel()
    descs[1].p, descs[1].m, descs[1].v, descs[1].n = X2.data_ptr(), Y2.data_ptr(), Z2.data_ptr(), X2.numel() // 32 * 32
    dims = (C.c_int64 * 2)(D, 32)
    mp = (C.c_void_p * 2)(marks.data_ptr(), None)
    N.check(lib.tt_adam_tables_sweep_marked(descs, dims, mp, 2, hyper2.data_ptr(), 0, N.stream()), "sweep_marked")
    tail = X2.numel() // 32 * 32  # (the second table's last 8 elements are no whole 32-wide row: swept by the plain call)
    N.check(lib.tt_adam_table_sweep(X2[tail:].data_ptr(), Y2[tail:].data_ptr(), Z2[tail:].data_ptr(), 1, X2.numel() - tail,
                                    hyper2.data_ptr(), N.stream()), "sweep tail")
    fj = (N.AdamFinishJob * 1)()
    j = fj[0]
    j.W, j.M, j.V, j.n_rows, j.dim, j.src, j.n_ids = W2.data_ptr(), M2.data_ptr(), V2.data_ptr(), n_rows, D, C.pointer(plan2.sources), n
    j.sorted_ids, j.perm, j.seg_begin, j.n_unique = (plan2.sorted_ids.data_ptr(), plan2.perm.data_ptr(),
                                                     plan2.seg_begin.data_ptr(), plan2.n_unique.data_ptr())
    j.side, j.side_bytes = None, 0
    N.check(lib.tt_adam_tables_finish(fj, 1, hyper2.data_ptr(), N.stream()), "tables_finish")
    torch.cuda.synchronize()
    for a, b, name in ((W, W2, "p"), (M, M2, "m"), (V, V2, "v"), (X, X2, "p'"), (Y, Y2, "m'"), (Z, Z2, "v'")):
        assert torch.equal(a, b), name
    assert not torch.equal(W.cpu(), W0)
    got = marks.cpu().numpy().view("uint32")
    want = torch.zeros(words * 32, dtype=torch.bool)
    want[ids.cpu()[ids.cpu() < n_rows]] = True
    import numpy as np
    assert np.array_equal(np.unpackbits(got.view("uint8"), bitorder="little").astype(bool), want.numpy())


@pytest.mark.parametrize("dims", [(128, 128), (128, 64), (50, 50), (128,), (128, 128, 128, 128, 128)])
def test_adam_merged_begin_and_finish_equal_the_separate_calls(T, dims):
    """tt_adam_begin_ids (advance + every table's plan-free stash) and tt_adam_tables_finish (every table's looked-up
    rows) in one launch each vs tt_adam_advance / tt_adam_table_stash_ids / tt_adam_table_finish per table: hyper, side
    buffers and tables bit for bit -- equal row widths (one launch), mixed widths, unaligned widths and five tables
    (the fall-backs inside the library)."""
    import ctypes as C
    ops, N = T
    lib = N.load()
    dev = torch.device(DEV)
    gen = torch.Generator().manual_seed(len(dims) * 1000 + dims[0])

    def run(merged):
        hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 4, 0, 0, 0], dtype=torch.float64, device=DEV)
        g2 = torch.Generator().manual_seed(17)
        tabs = []
        for k, D in enumerate(dims):
            n_rows, n = 500 + 37 * k, 300 + 11 * k
            W = torch.randn(n_rows, D, generator=g2).to(DEV)
            M, V = (torch.randn(n_rows, D, generator=g2) * 0.01).to(DEV), (torch.rand(n_rows, D, generator=g2) * 1e-4).to(DEV)
            ids = torch.randint(0, n_rows, (n,), generator=g2)
            ids[:5] = ids[5:10]
            ids[10] = n_rows + 3  # an id outside the block parks zeros and is skipped by the finish
            rows = (torch.randn(n, D, generator=g2) * 0.01).to(DEV)
            side = torch.full((lib.tt_adam_table_workspace_bytes(n, D) + 64,), 7, dtype=torch.uint8, device=DEV)
            tabs.append([W, M, V, ids.to(DEV), rows, side, n_rows, n, D])
        if merged:
            jobs = (N.AdamStashJob * len(tabs))()
            for j, (W, M, V, ids, rows, side, n_rows, n, D) in zip(jobs, tabs):
                j.W, j.M, j.V, j.n_rows, j.dim, j.ids, j.n_ids = W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, D, ids.data_ptr(), n
                j.side, j.side_bytes = side.data_ptr(), side.numel()
            N.check(lib.tt_adam_begin_ids(hyper.data_ptr(), None, 0, jobs, len(tabs), N.stream()), "begin_ids")
        else:
            N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "advance")
            for W, M, V, ids, rows, side, n_rows, n, D in tabs:
                N.check(lib.tt_adam_table_stash_ids(W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, D, ids.data_ptr(), n,
                                                    side.data_ptr(), side.numel(), N.stream()), "stash_ids")
        sides = [t[5].clone() for t in tabs]
        plans = []
        for W, M, V, ids, rows, side, n_rows, n, D in tabs:
            plan = ops.RowPlan([ids], n_rows + 8)  # rows beyond n_rows sort last; the finish skips them
            plan.attach([rows])
            plans.append(plan)
        if merged:
            fj = (N.AdamFinishJob * len(tabs))()
            for j, plan, (W, M, V, ids, rows, side, n_rows, n, D) in zip(fj, plans, tabs):
                j.W, j.M, j.V, j.n_rows, j.dim, j.src, j.n_ids = W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, D, C.pointer(plan.sources), n
                j.sorted_ids, j.perm, j.seg_begin, j.n_unique = (plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                                                 plan.seg_begin.data_ptr(), plan.n_unique.data_ptr())
                j.side, j.side_bytes = side.data_ptr(), side.numel()
            N.check(lib.tt_adam_tables_finish(fj, len(tabs), hyper.data_ptr(), N.stream()), "tables_finish")
        else:
            for plan, (W, M, V, ids, rows, side, n_rows, n, D) in zip(plans, tabs):
                N.check(lib.tt_adam_table_finish(W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, D, hyper.data_ptr(),
                                                 C.byref(plan.sources), n, plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                                 plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), side.data_ptr(),
                                                 side.numel(), N.stream()), "finish")
        torch.cuda.synchronize()
        return hyper, sides, [(t[0], t[1], t[2]) for t in tabs]

    ha, sa, ta = run(True)
    hb, sb, tb = run(False)
    assert torch.equal(ha, hb) and float(ha[4]) == 5.0
    for x, y in zip(sa, sb):
        assert torch.equal(x, y)
    for (w1, m1, v1), (w2, m2, v2) in zip(ta, tb):
        assert torch.equal(w1, w2) and torch.equal(m1, m2) and torch.equal(v1, v2)
        assert not torch.equal(w1, torch.zeros_like(w1))


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,D,heads", [(3, 50, 128, 4), (2, 7, 40, 4), (1, 3, 2, 1), (2, 128, 64, 4), (5, 16, 40, 4),
                                          (2, 20, 256, 2), (2, 9, 200, 2),
                                         (4, 64, 64, 4), (2, 33, 128, 2), (3, 1, 128, 4), (9, 50, 64, 2)])
def test_attention_forward_backward(T, B, H, D, heads):
    ops, N = T
    lib = N.load()
    qkv = (g((B * H, 3 * D), 91) * 0.7).requires_grad_(True)
    dh = D // heads

    def split(t):
        return t.reshape(B, H, heads, dh).permute(0, 2, 1, 3)

    q, k, v = split(qkv[:, :D]), split(qkv[:, D:2 * D]), split(qkv[:, 2 * D:])
    s = (q / math.sqrt(dh)) @ k.transpose(-1, -2)
    ctx_ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * H, D)
    cot = g((B * H, D), 92)
    (ctx_ref * cot).sum().backward()
    qd = qkv.detach().to(DEV)
    ctx, lse = ops._attn_fwd(qd, B, H, D, heads)
    assert torch.allclose(ctx.cpu(), ctx_ref.detach(), atol=2e-6, rtol=1e-5)
    assert torch.allclose(lse.cpu(), torch.logsumexp(s, -1).detach(), atol=1e-5)
    d_qkv = torch.empty_like(qd)
    cot_d = cot.to(DEV)
    N.check(lib.tt_attn_bwd(qd.data_ptr(), ctx.data_ptr(), lse.data_ptr(), cot_d.data_ptr(), B, H, D, heads,
                            d_qkv.data_ptr(), N.stream()), "attn_bwd")
    assert torch.allclose(d_qkv.cpu(), qkv.grad, atol=2e-6 * float(qkv.grad.abs().max()) + 1e-7, rtol=2e-4)


@pytest.mark.parametrize("B,H,D,heads", [(9, 50, 128, 4), (5, 64, 64, 4), (3, 1, 32, 2), (70, 7, 48, 3), (4, 3, 4, 1),
                                         (300, 50, 128, 4), (33, 10, 100, 5), (2, 64, 128, 16)])
def test_encoder_last_layer_without_kv_projection(T, B, H, D, heads):
    """tt_enc_last_fwd / _bwd (csrc/encoder_last.hip): the encoder's last attention layer, consumed at row 0 only
    (ref:src/user_history_encoder.py:103-116), with K / V never projected -- against the oracle's full layer
    (oracle/cpu_ref.self_attention_layer, every position, row 0 taken) and torch autograd through it: output,
    input gradient, and all four parameter gradients (the K third of the in-projection bias: zero both ways).
    (The form that reads the PREVIOUS layer's context through composed weights is the caller's arithmetic:
    tests/test_gpu_models.py::test_encoder_matches_reference covers it at L = 3.)"""
    ops, N = T
    lib = N.load()
    assert lib.tt_enc_last_supported(H, D, heads) == 1
    x = g((B, H, D), 301).requires_grad_(True)
    w_in = (g((3 * D, D), 302) * (1.0 / math.sqrt(D))).requires_grad_(True)
    b_in = (g((3 * D,), 303) * 0.1).requires_grad_(True)
    w_out = (g((D, D), 304) * (1.0 / math.sqrt(D))).requires_grad_(True)
    b_out = (g((D,), 305) * 0.1).requires_grad_(True)
    ref = R.self_attention_layer(x, w_in, b_in, w_out, b_out, heads)[:, 0, :]
    cot = g((B, 2 * D), 306)[:, :D]  # strided rows, like out[:, 0, :] of the [B, 2, D] encoder output
    (ref * cot).sum().backward()
    xd = x.detach().reshape(B * H, D).to(DEV)
    wi, bi, wo, bo = (t.detach().to(DEV) for t in (w_in, b_in, w_out, b_out))
    out = torch.zeros(B, 2 * D, device=DEV)
    q0, ctx0 = torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
    tq, xbar = torch.empty(B, heads, D, device=DEV), torch.empty(B, heads, D, device=DEV)
    probs = torch.empty(B, heads, H, device=DEV)
    N.check(lib.tt_enc_last_fwd(xd.data_ptr(), B, H, D, heads, wi.data_ptr(), bi.data_ptr(), wo.data_ptr(), bo.data_ptr(),
                                out.data_ptr(), 2 * D, q0.data_ptr(), tq.data_ptr(), probs.data_ptr(), xbar.data_ptr(),
                                ctx0.data_ptr(), N.stream()), "enc_last_fwd")
    assert torch.allclose(out[:, :D].cpu(), ref.detach(), atol=4e-6, rtol=1e-5)
    assert float(out[:, D:].abs().max()) == 0.0  # the other half of each row is not touched
    assert torch.allclose(probs.sum(-1).cpu(), torch.ones(B, heads), atol=1e-5)
    cd = torch.zeros(B, 2 * D, device=DEV)
    cd[:, :D] = cot.to(DEV)
    dx = torch.empty(B * H, D, device=DEV)
    dWi, dbi = torch.empty(3 * D, D, device=DEV), torch.empty(3 * D, device=DEV)
    dWo, dbo = torch.empty(D, D, device=DEV), torch.empty(D, device=DEV)
    nb = lib.tt_enc_last_bwd_workspace_bytes(B, H, D, heads)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    for _ in range(2):  # twice: the partial buffers are overwritten, not accumulated into
        N.check(lib.tt_enc_last_bwd(xd.data_ptr(), B, H, D, heads, wi.data_ptr(), wo.data_ptr(), cd.data_ptr(), 2 * D,
                                    q0.data_ptr(), tq.data_ptr(), probs.data_ptr(), xbar.data_ptr(), ctx0.data_ptr(),
                                    dx.data_ptr(), dWi.data_ptr(), dbi.data_ptr(), dWo.data_ptr(), dbo.data_ptr(),
                                    ws.data_ptr(), nb, N.stream()), "enc_last_bwd")

    def close(got, want, name):
        tol = 4e-6 * float(want.abs().max()) + 1e-8
        assert torch.allclose(got.cpu().reshape(want.shape), want, atol=tol, rtol=2e-4), (name, float((got.cpu().reshape(want.shape) - want).abs().max()), tol)

    close(dx, x.grad, "dx")
    close(dWo, w_out.grad, "dW_out")
    close(dbo, b_out.grad, "db_out")
    close(dWi, w_in.grad, "dW_in")
    assert float(dbi[D:2 * D].abs().max()) == 0.0 and float(b_in.grad[D:2 * D].abs().max()) < 1e-5 * float(b_in.grad.abs().max()) + 1e-7
    keep = torch.ones(3 * D, dtype=torch.bool)
    keep[D:2 * D] = False
    close(dbi[keep.to(DEV)], b_in.grad[keep], "db_in")


@pytest.mark.parametrize("D,H,B", [(128, 50, 9), (50, 7, 4)])
def test_hist_embed_pool(T, D, H, B):
    ops, N = T
    lib = N.load()
    table = g((400, D), 95)
    ids = torch.from_numpy(fg.uniform_ids((B, H), 400, 96))
    pe = R.positional_table(H, D)
    x = torch.empty(B * H, D, device=DEV)
    out = torch.zeros(B, 2, D, device=DEV)
    table_d, ids_d, pe_d = table.to(DEV), ids.to(DEV), pe.to(DEV)  # keep the device copies alive
    N.check(lib.tt_hist_embed_pool(table_d.data_ptr(), 400, D, ids_d.data_ptr(), B, H,
                                   pe_d.data_ptr(), x.data_ptr(), out[:, 1, :].data_ptr(), 2 * D,
                                   N.oob.flag(torch.device(DEV)).data_ptr(), N.stream()), "hist_embed_pool")
    emb = table[ids]
    assert torch.allclose(x.cpu().view(B, H, D), emb + pe, atol=1e-7)
    assert torch.allclose(out[:, 1, :].cpu(), emb.mean(1), atol=1e-6)
    assert float(out[:, 0, :].abs().max()) == 0.0


# ------------------------------------------------------------------ tall-M weights-stationary GEMM
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("M,Nn,K", [(20000, 384, 128), (16389, 130, 100), (16384, 128, 256), (17000, 50, 200),
                                    (16500, 128, 384), (16390, 70, 300),
                                    (40000, 128, 32), (16400, 257, 64),
                                    # N = 128 exactly, K in {128, 256, 384}: gemm_ws16.hip (16 columns per wave, three-stage ring);
                                    # ragged last stages (rows % 32 / 48 / 96 != 0) and a single-stage problem per workgroup
                                    (204800, 128, 128), (16385, 128, 128), (20001, 128, 256), (33333, 128, 384),
                                    (204800, 384, 128), (17777, 384, 128), (30001, 256, 128)])
def test_gemm_weights_stationary_path(T, layout, M, Nn, K):
    """M >= 16384 and K <= 256 route NT / NN products to gemm_ws.hip (LDS-DMA when K is 32/64/128/256
    and rows are aligned, register staging otherwise); epilogues, strides and accumulate included."""
    ops, N = T
    A = g((M, K), 201)
    W = g((Nn, K) if layout == 0 else (K, Nn), 202)
    bias = g((Nn,), 203)
    ref = A.double() @ (W.t() if layout == 0 else W).double() + bias.double()
    Ad, Wd = A.to(DEV), W.to(DEV)
    big = torch.zeros(M, Nn + 8, device=DEV)
    out = big[:, 4:4 + Nn]  # strided, 16-B misaligned destination
    ops.gemm(layout, Ad, Wd, out, M, Nn, K, bias=bias.to(DEV))
    scale = float(ref.abs().max())
    assert float((out.cpu().double() - ref).abs().max()) < 3e-6 * math.sqrt(K) * scale
    assert float(big[:, :4].abs().max()) == 0.0 and float(big[:, 4 + Nn:].abs().max()) == 0.0
    aux = g((M, Nn), 204).to(DEV)
    out2 = torch.empty(M, Nn, device=DEV)
    ops.gemm(layout, Ad, Wd, out2, M, Nn, K, bias=bias.to(DEV), epilogue=N.TT_EPI_RELU_MASK, aux=aux)
    assert torch.allclose(out2.cpu().double(), ref * (aux.cpu() > 0), atol=3e-6 * math.sqrt(K) * scale)
    ops.gemm(layout, Ad, Wd, out2, M, Nn, K, epilogue=N.TT_EPI_RELU, accumulate=True)
    want = ref * (aux.cpu() > 0) + (ref - bias.double()).clamp(min=0)
    assert torch.allclose(out2.cpu().double(), want, atol=6e-6 * math.sqrt(K) * scale)


@pytest.mark.parametrize("B,D,F", [(8192, 128, 8), (100, 128, 64), (333, 64, 20), (65, 32, 1), (1, 128, 33)])
def test_tower_weight_gradients_kernel_vs_fp64(T, B, D, F):
    """tt_tower_bwd_weights (block partials in one launch + one reduce) on its own: dW3 = dy^T tin, dW2 = d_f^T h,
    dW1 = dh^T x and the three bias sums against float64 products, incl. a ragged last 64-row block, one row, F = 64;
    and bit-identical from call to call (the reduce adds the partials in block order)."""
    ops, N = T
    g = torch.Generator().manual_seed(B * 7 + D)
    dy, tin, d_f = torch.randn(B, D, generator=g), torch.randn(B, 2 * D, generator=g), torch.randn(B, D, generator=g)
    h, dh, x = torch.randn(B, 256, generator=g).relu(), torch.randn(B, 256, generator=g), torch.randn(B, F, generator=g)
    dev = [t.to(DEV) for t in (dy, tin, d_f, h, dh, x)]
    got = ops.tower_weight_grads(*dev)
    again = ops.tower_weight_grads(*dev)
    want = ((dh.double().t() @ x.double()), dh.double().sum(0), (d_f.double().t() @ h.double()), d_f.double().sum(0),
            (dy.double().t() @ tin.double()), dy.double().sum(0))
    for a, b, w in zip(got, again, want):
        assert torch.equal(a, b)
        assert torch.allclose(a.cpu().double(), w, rtol=1e-5, atol=2e-5 * math.sqrt(B))


@pytest.mark.parametrize("B,D,F,n_rows", [(4096, 128, 8, 100_000), (100, 128, 8, 50), (333, 64, 20, 1000), (65, 32, 8, 200),
                                          (1, 128, 33, 7), (8192, 128, 8, 5000)])  # > 4096 rows: the 64-row-per-workgroup form
def test_fused_tower_with_third_input_block_forward_and_backward(T, B, D, F, n_rows):
    """tt_tower_fwd_x / tt_tower_bwd_data_x / tt_tower_bwd_weights_x: the history model's user tower
    [ id | MLP | recent | mean ] -> Linear(4D -> D) (ref:src/two_tower_with_user_history_encoder.py:81-83,85-122) as one
    kernel per direction, against the same lines in float64 torch on the CPU: output, all six parameter gradients, the
    embedding-row gradients and the gradient that flows back into the encoder summary; ragged last block, one row,
    duplicate ids; the extra block as a strided view; bit-identical from call to call."""
    ops, N = T
    gen = torch.Generator().manual_seed(3 * B + D)
    p = {"emb": torch.randn(n_rows, D, generator=gen), "W1": torch.randn(256, F, generator=gen) * 0.3,
         "b1": torch.randn(256, generator=gen) * 0.1, "W2": torch.randn(D, 256, generator=gen) * 0.06,
         "b2": torch.randn(D, generator=gen) * 0.1, "W3": torch.randn(D, 4 * D, generator=gen) * 0.05,
         "b3": torch.randn(D, generator=gen) * 0.1, "extra": torch.randn(B, 2 * D, generator=gen)}
    ids = torch.randint(0, n_rows, (B,), generator=gen)
    if B > 8:
        ids[:4] = ids[4:8]
    feats, cot = torch.randn(B, F, generator=gen), torch.randn(B, D, generator=gen)
    ref = {k: v.double().requires_grad_(True) for k, v in p.items()}
    hid = torch.relu(feats.double() @ ref["W1"].t() + ref["b1"])
    tin = torch.cat([ref["emb"][ids], hid @ ref["W2"].t() + ref["b2"], ref["extra"]], dim=1)
    want = tin @ ref["W3"].t() + ref["b3"]
    (want * cot.double()).sum().backward()
    outs = []
    for rep in range(2):
        dl = {k: v.clone().to(DEV).requires_grad_(True) for k, v in p.items()}
        wide = torch.zeros(B, 2 * D + 8, device=DEV)
        wide[:, 4:4 + 2 * D] = dl["extra"].detach()
        extra = dl["extra"] if rep == 0 else wide[:, 4:4 + 2 * D].requires_grad_(False)  # rep 1: 16-B aligned strided rows
        assert ops.fused_tower_supported(dl["emb"], feats.to(DEV), dl["W1"], dl["W2"], dl["W3"], extra_width=2 * D)
        y = ops.FusedTower.apply(dl["emb"], ids.to(DEV), feats.to(DEV), dl["W1"], dl["b1"], dl["W2"], dl["b2"], dl["W3"], dl["b3"],
                                 extra)
        assert torch.allclose(y.cpu().double(), want.detach(), atol=2e-5, rtol=1e-5)
        (y * cot.to(DEV)).sum().backward()
        outs.append({k: dl[k].grad.clone() for k in p if dl[k].grad is not None})
        for k, g in outs[-1].items():
            g_ref = ref[k].grad
            assert torch.allclose(g.cpu().double(), g_ref, atol=1e-5 * float(g_ref.abs().max()) + 1e-8, rtol=2e-4), k
    assert "extra" in outs[0] and "W3" in outs[1]
    for k in outs[1]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("B,D,F,n_rows", [(8192, 128, 8, 100_000), (100, 128, 8, 50), (333, 64, 20, 1000), (64, 32, 8, 200),
                                          (1, 128, 33, 7), (4200, 64, 8, 5000)])  # > 4096 rows: the 64-row-per-workgroup form
def test_fused_tower_matches_oracle_forward_and_backward(T, B, D, F, n_rows):
    """tt_tower_fwd / tt_tower_bwd_data (one launch per direction: lookup + feature MLP + cat + tower Linear,
    ref:src/two_tower_base_retrieval.py:129-219) against the CPU oracle's item tower: output, every parameter gradient,
    and the embedding-row gradients (dense form), incl. duplicate ids and a ragged last 64-row block."""
    ops, N = T
    gen = torch.Generator().manual_seed(B + D)
    p = {"item_id_embedding_arch.weight": torch.randn(n_rows, D, generator=gen),
         "item_features_arch.0.weight": torch.randn(256, F, generator=gen) * 0.3, "item_features_arch.0.bias": torch.randn(256, generator=gen) * 0.1,
         "item_features_arch.2.weight": torch.randn(D, 256, generator=gen) * 0.06, "item_features_arch.2.bias": torch.randn(D, generator=gen) * 0.1,
         "item_tower_arch.weight": torch.randn(D, 2 * D, generator=gen) * 0.06, "item_tower_arch.bias": torch.randn(D, generator=gen) * 0.1}
    ids = torch.randint(0, n_rows, (B,), generator=gen)
    if B > 8:
        ids[:4] = ids[4:8]
    feats = torch.randn(B, F, generator=gen)
    cot = torch.randn(B, D, generator=gen)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    want = R.item_embeddings(leaves, ids, feats)
    (want * cot).sum().backward()
    dl = {k: v.clone().to(DEV).requires_grad_(True) for k, v in p.items()}
    assert ops.fused_tower_supported(dl["item_id_embedding_arch.weight"], feats.to(DEV), dl["item_features_arch.0.weight"],
                                     dl["item_features_arch.2.weight"], dl["item_tower_arch.weight"])
    y = ops.FusedTower.apply(dl["item_id_embedding_arch.weight"], ids.to(DEV), feats.to(DEV), dl["item_features_arch.0.weight"],
                             dl["item_features_arch.0.bias"], dl["item_features_arch.2.weight"], dl["item_features_arch.2.bias"],
                             dl["item_tower_arch.weight"], dl["item_tower_arch.bias"])
    assert torch.allclose(y.cpu(), want.detach(), atol=1e-5, rtol=1e-5)
    (y * cot.to(DEV)).sum().backward()
    for k in p:
        g_ref, g = leaves[k].grad, dl[k].grad.cpu()
        assert torch.allclose(g, g_ref, atol=1e-5 * float(g_ref.abs().max()) + 1e-8, rtol=2e-4), k


@pytest.mark.parametrize("B,D,Fu,Fi", [(300, 64, 5, 20), (8192, 128, 8, 8), (4096, 32, 16, 3), (33, 128, 8, 8)])
def test_tower_pair_launches_are_the_single_tower_kernels_bit_for_bit(B, D, Fu, Fi):
    """tt_tower_*_pair (both towers of the base model per launch: ops.FusedTowerPair) against two ops.FusedTower calls:
    embeddings and every gradient -- table rows, the six weight / bias tensors of each tower -- identical bits."""
    import torch.nn as nn
    from two_tower_models_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + D)

    def tower(n_rows, F):
        P = lambda *s: nn.Parameter((torch.randn(*s, generator=g) * 0.2).to(dev))
        return [P(n_rows, D), None, None, P(256, F), P(256), P(D, 256), P(D), P(D, 2 * D), P(D)]

    u, i = tower(1000, Fu), tower(700, Fi)
    u[1], u[2] = torch.randint(0, 1000, (B,), generator=g).to(dev), torch.randn(B, Fu, generator=g).to(dev)
    i[1], i[2] = torch.randint(0, 700, (B,), generator=g).to(dev), torch.randn(B, Fi, generator=g).to(dev)
    gu, gi = torch.randn(B, D, generator=g).to(dev), torch.randn(B, D, generator=g).to(dev)
    params = [t for t in u + i if isinstance(t, nn.Parameter)]

    def grads():
        out = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        return out

    yu, yi = ops.FusedTower.apply(*u), ops.FusedTower.apply(*i)
    torch.autograd.backward([yu, yi], [gu, gi])
    want = grads()
    pu, pi = ops.FusedTowerPair.apply(*u, *i)
    assert torch.equal(pu, yu) and torch.equal(pi, yi)
    torch.autograd.backward([pu, pi], [gu, gi])
    got = grads()
    assert all(torch.equal(a, b) for a, b in zip(got, want))


def test_row_plans_of_several_tables_in_one_launch_match_the_single_plans():
    """tt_rowgrad_plan_jobs (one workgroup per id list: ops.RowPlan.build_many) against tt_rowgrad_plan list by list:
    sorted ids, permutation, run starts and run count identical -- incl. a list of one id, heavy duplication, the longest
    list the one-workgroup sort takes (10 240) and a mix with a longer list (falls back to one plan after the other)."""
    from two_tower_models_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    cases = [(8192, 10_000_000), (8192, 1_000_000), (1, 7), (300, 50), (10240, (1 << 31) - 5)]
    lists = [torch.randint(0, rows, (n,), generator=g).to(dev) for n, rows in cases]
    single = [ops.RowPlan([ids], rows) for ids, (_, rows) in zip(lists, cases)]

    def same(a, b):
        nu = int(a.n_unique.item())
        return (nu == int(b.n_unique.item()) and torch.equal(a.sorted_ids, b.sorted_ids) and torch.equal(a.perm, b.perm)
                and torch.equal(a.seg_begin[: nu + 1], b.seg_begin[: nu + 1]))

    for lo in (0, 2):  # (at most 4 jobs per launch)
        many = [ops.RowPlan([ids], rows, defer=True) for ids, (_, rows) in zip(lists[lo:lo + 3], cases[lo:lo + 3])]
        ops.RowPlan.build_many(many)
        assert all(same(a, b) for a, b in zip(many, single[lo:lo + 3]))
    long_ids = torch.randint(0, 1000, (20000,), generator=g).to(dev)
    mixed = [ops.RowPlan([lists[0]], cases[0][1], defer=True), ops.RowPlan([long_ids], 1000, slot="plan_b", defer=True)]
    ops.RowPlan.build_many(mixed)
    assert same(mixed[0], single[0]) and same(mixed[1], ops.RowPlan([long_ids], 1000, slot="plan_c"))
