"""BASELINE-size checks on MI355X (`pytest -m gpu`).  The oracle cannot run these sizes in
seconds, so each kernel is checked through size-independent properties plus oracle values
on a SAMPLE of rows (rows / samples / queries are independent in every kernel here)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def T():
    from two_tower_models_amd import _native as N
    from two_tower_models_amd import ops
    N.load()
    return ops, N


def test_inbatch_ce_b8192_d128(T):
    ops, N = T
    B, D = 8192, 128
    g = torch.Generator(device="cpu").manual_seed(3)
    U = (torch.randn(B, D, generator=g) * 0.3)
    I = (torch.randn(B, D, generator=g) * 0.3)
    coef = torch.rand(B, generator=g) / B
    Ud, Id = U.to(DEV).requires_grad_(True), I.to(DEV).requires_grad_(True)
    ce = ops.InBatchSoftmaxCE.apply(Ud, Id, 0)
    (ce * coef.to(DEV)).sum().backward()
    rows = torch.arange(0, B, 131)
    S = U[rows].double() @ I.double().t()                       # oracle rows, fp64
    lse = torch.logsumexp(S, 1)
    assert torch.allclose(ce.cpu()[rows].double(), lse - S[torch.arange(len(rows)), rows], atol=2e-5)
    P = torch.exp(S - lse[:, None])
    P[torch.arange(len(rows)), rows] -= 1.0
    dU_ref = (P * coef[rows, None].double()) @ I.double()
    assert torch.allclose(Ud.grad.cpu()[rows].double(), dU_ref, atol=1e-9, rtol=2e-4)
    # sum_j dI[j] = U^T (G 1) = 0 (softmax rows sum to one): bounded by rounding
    assert float(Id.grad.sum(0).abs().max()) < 1e-6
    # exact linearity in coef: doubling it doubles both gradients bit for bit
    U2, I2 = U.to(DEV).requires_grad_(True), I.to(DEV).requires_grad_(True)
    (ops.InBatchSoftmaxCE.apply(U2, I2, 0) * (2 * coef).to(DEV)).sum().backward()
    assert torch.equal(U2.grad, 2 * Ud.grad) and torch.equal(I2.grad, 2 * Id.grad)
    # global-negative layout (8 ranks' items, this rank's positives at 3*B): same rows
    I_all = torch.cat([torch.randn(3 * B, D, generator=g) * 0.3, I, torch.randn(4 * B, D, generator=g) * 0.3])
    ce8 = ops.InBatchSoftmaxCE.apply(U.to(DEV), I_all.to(DEV), 3 * B)
    S8 = U[rows].double() @ I_all.double().t()
    want = torch.logsumexp(S8, 1) - S8[torch.arange(len(rows)), rows + 3 * B]
    assert torch.allclose(ce8.cpu()[rows].double(), want, atol=2e-5)


def test_inbatch_ce_8_rank_shape_kept_logits(T):
    """One rank's logits at 8 GPUs (8192 users x 65 536 items, positives at 3*B): the kept-logits pair
    (tt_inbatch_ce_fwd_du_keep / tt_inbatch_ce_bwd_kept, the default of ops.InBatchSoftmaxCE for N >= 4 M)
    against the recomputing pair, and both against size-independent properties / sampled fp64 rows."""
    ops, N = T
    B, D, W = 8192, 128, 8
    g = torch.Generator(device="cpu").manual_seed(5)
    U = torch.randn(B, D, generator=g) * 0.3
    I_all = torch.randn(W * B, D, generator=g) * 0.3
    coef = torch.rand(B, generator=g) / (W * B)
    res = []
    for keep in (True, False):
        Ud, Id = U.to(DEV).requires_grad_(True), I_all.to(DEV).requires_grad_(True)
        ce = ops.InBatchSoftmaxCE.apply(Ud, Id, 3 * B, keep)
        (ce * coef.to(DEV)).sum().backward()
        res.append((ce.detach(), Ud.grad, Id.grad))
    # same forward statistics and user gradient bit for bit; the item gradient comes from the same
    # arithmetic on stored instead of recomputed logits
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.allclose(res[0][2], res[1][2], atol=1e-7 * float(res[1][2].abs().max()), rtol=1e-5)
    ce, dU, dI = res[0]
    assert float(dI.sum(0).abs().max()) < 1e-6  # softmax rows sum to one
    rows = torch.arange(0, B, 257)
    S = U[rows].double() @ I_all.double().t()
    lse = torch.logsumexp(S, 1)
    assert torch.allclose(ce.cpu()[rows].double(), lse - S[torch.arange(len(rows)), rows + 3 * B], atol=2e-5)
    # sampled item rows: dI[j] = sum_i coef_i (p_ij - [j == i + 3B]) U_i needs every user -> fp64 on the GPU
    items = torch.tensor([0, 12345, 3 * B, 3 * B + 4097, 4 * B - 1, W * B - 1])
    Sd = U.to(DEV).double() @ I_all[items].to(DEV).double().t()              # [B, 6]
    lse_all = torch.logsumexp(U.to(DEV).double() @ I_all.to(DEV).double().t(), 1)  # [B]
    G = torch.exp(Sd - lse_all[:, None])
    for k, j in enumerate(items.tolist()):
        if 3 * B <= j < 4 * B:
            G[j - 3 * B, k] -= 1.0
    ref = (G * coef.to(DEV).double()[:, None]).t() @ U.to(DEV).double()
    got = dI[items.to(DEV)].double()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-12


def test_row_plan_420k_ids_over_10m_rows(T):
    ops, N = T
    n, n_rows = 8192 * 51, 10_000_000
    ids = torch.randint(0, n_rows, (n,), generator=torch.Generator().manual_seed(5)).to(DEV)
    ids[:1000] = ids[1000:2000]  # guaranteed duplicates
    plan = ops.RowPlan([ids], n_rows)
    srt = plan.sorted_ids.long()
    assert bool((srt[1:] >= srt[:-1]).all())
    assert torch.equal(ids[plan.perm.long()], srt)                     # perm realises the sort
    assert torch.equal(torch.sort(plan.perm.long()).values, torch.arange(n, device=DEV))
    same = srt[1:] == srt[:-1]
    assert bool((plan.perm[1:][same] > plan.perm[:-1][same]).all())    # stable
    U = int(plan.n_unique.item())
    assert U == int(torch.unique(ids).numel())
    seg = plan.seg_begin[: U + 1].long()
    assert int(seg[0]) == 0 and int(seg[U]) == n and bool((seg[1:] > seg[:-1]).all())
    assert bool((srt[seg[1:U]] != srt[seg[1:U] - 1]).all())


def test_adam_table_2m_rows_untouched_rows_and_sample(T):
    from oracle import cpu_ref as R
    ops, N = T
    lib = N.load()
    n_rows, D, n = 2_000_000, 128, 8192
    g = torch.Generator().manual_seed(7)
    W = torch.randn(n_rows, D, generator=g)
    Wd, Md, Vd = W.to(DEV), torch.zeros(n_rows, D, device=DEV), torch.zeros(n_rows, D, device=DEV)
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0, 0, 0, 0], dtype=torch.float64, device=DEV)
    steps = []
    for _ in (1, 2):
        ids = torch.randint(0, n_rows, (n,), generator=g)
        ids[:64] = ids[64:128]  # duplicates inside a step
        steps.append((ids, torch.randn(n, D, generator=g) * 0.01))
    steps[1][0][:32] = steps[0][0][:32]  # rows touched in BOTH steps
    touched = set()
    for step, (ids, rows) in enumerate(steps, start=1):
        N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "adv")
        plan = ops.RowPlan.from_grads([ops.RowGrad(ids.to(DEV), rows.to(DEV))], n_rows)
        wsp, wsn = ops._ws(torch.device(DEV), lib.tt_adam_table_workspace_bytes(plan.n, D), "adam_side")
        N.check(lib.tt_adam_table(Wd.data_ptr(), Md.data_ptr(), Vd.data_ptr(), n_rows, D, hyper.data_ptr(),
                                  C.byref(plan.sources), plan.n, plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                  plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), wsp, wsn, N.stream()), "adam")
        touched |= set(ids.tolist())
    # oracle on a sample of rows, each followed through BOTH steps with that step's summed gradient
    sample = set(steps[0][0][:200].tolist()) | set(steps[1][0][:200].tolist())
    ref = {}
    for k in sample:
        p, m, v = W[k].clone(), torch.zeros(D), torch.zeros(D)
        for step, (ids, rows) in enumerate(steps, start=1):
            hit = ids == k
            R.adam_update(p, rows[hit].sum(0) if bool(hit.any()) else torch.zeros(D), m, v, step)
        ref[k] = (p, m, v)
    got = Wd.cpu()
    for k, (p, m, v) in ref.items():
        assert torch.allclose(got[k], p, atol=3e-6), k
    mask = torch.ones(n_rows, dtype=torch.bool)
    mask[torch.tensor(sorted(touched))] = False
    assert torch.equal(got[mask], W[mask])                  # never-looked-up rows: bit-identical
    assert float(Md.cpu()[mask].abs().max()) == 0.0


@pytest.mark.parametrize("bf16", [False, True])
def test_mips_1m_corpus_selection_is_optimal(T, bf16):
    import two_tower_models_amd as A
    C_, D, B, K = 1_000_000, 128, 96, 1000
    g = torch.Generator(device=DEV).manual_seed(11)
    corpus = torch.randn(C_, D, device=DEV, generator=g)
    q = torch.randn(B, D, device=DEV, generator=g)
    m = A.BaselineMIPSModule(corpus_size=8, embedding_dim=D)
    m.corpus = corpus.clone()
    m.corpus_size = C_
    if bf16:
        m.use_bf16_storage()
        corpus = m.corpus.float()
        q_eff = q.to(torch.bfloat16).float()
    else:
        q_eff = q
    idx, sc = m.search(q, K)
    assert idx.shape == (B, K) and bool(((idx >= 0) & (idx < C_)).all())
    assert bool((sc[:, 1:] <= sc[:, :-1]).all())                              # sorted descending
    assert all(len(set(r.tolist())) == K for r in idx[:8].cpu())              # no duplicates
    full = q_eff.double() @ corpus.double().t()                               # [B, C] fp64 checker
    picked = torch.gather(full, 1, idx)
    assert torch.allclose(picked, sc.double(), atol=5e-4)
    kth = torch.topk(full, K, dim=1).values[:, -1]
    assert bool((picked.min(1).values >= kth - 5e-4).all())                   # nothing better was left out
    ref_idx = torch.topk(full, K, dim=1).indices
    overlap = np.mean([len(set(a.tolist()) & set(b.tolist())) / K for a, b in zip(idx.cpu(), ref_idx.cpu())])
    assert overlap > 0.998


def test_history_encoder_b4096_h50_sampled(T):
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    torch.manual_seed(0)
    B, H, D = 4096, 50, 128
    enc = A.UserHistoryEncoder(D, H, 4, 3, True)
    params = {k: v.clone() for k, v in enc.state_dict().items()}
    enc = enc.to(DEV)
    x = torch.randn(B, H, D)
    y = enc(x.to(DEV))
    rows = torch.arange(0, B, 517)
    want = R.history_encoder_forward(x[rows], R.encoder_layers_from_params(params, prefix=""), 4,
                                     R.positional_table(H, D))
    assert torch.allclose(y.cpu()[rows], want, atol=2e-5, rtol=1e-4)
