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


@pytest.mark.parametrize("keep", [False, True])
def test_split_fp16_ce_pair_8_rank_shape(T, keep, monkeypatch):
    """EXPLORATORY path, full size: the split-fp16 logits pair (csrc/ce_f16x2.hip, TT_CE_F16X2) through
    ops.InBatchSoftmaxCE at one rank's 8-GPU shape -- 8192 users x 65 536 items, positives at 3*B -- in its default form
    (no logits buffer, `keep` False) and its kept-logits form, with the SAME sampled-float64 assertions and tolerances as
    the fp32-MFMA pair (test_inbatch_ce_8_rank_shape_kept_logits above)."""
    ops, N = T
    B, D, W = 8192, 128, 8
    assert N.load().tt_ce16_supported(B, W * B, D) == 1
    monkeypatch.setattr(ops, "_CE_F16X2", True)
    monkeypatch.setattr(ops, "_CE16_KEEP", bool(keep))
    calls = []
    lib = N.load()
    which = "tt_ce16_bwd_kept" if keep else "tt_ce16_bwd_recompute"
    real = getattr(lib, which)
    monkeypatch.setattr(lib, which, lambda *a: (calls.append(1), real(*a))[1])
    g = torch.Generator(device="cpu").manual_seed(5)
    U = torch.randn(B, D, generator=g) * 0.3
    I_all = torch.randn(W * B, D, generator=g) * 0.3
    coef = torch.rand(B, generator=g) / (W * B)
    Ud, Id = U.to(DEV).requires_grad_(True), I_all.to(DEV).requires_grad_(True)
    ce = ops.InBatchSoftmaxCE.apply(Ud, Id, 3 * B)
    (ce * coef.to(DEV)).sum().backward()
    assert calls == [1]  # the pair is what ran
    ce, dU, dI = ce.detach(), Ud.grad, Id.grad
    assert float(dI.sum(0).abs().max()) < 1e-6  # softmax rows sum to one
    rows = torch.arange(0, B, 257)
    S = U[rows].double() @ I_all.double().t()
    lse = torch.logsumexp(S, 1)
    assert torch.allclose(ce.cpu()[rows].double(), lse - S[torch.arange(len(rows)), rows + 3 * B], atol=2e-5)
    # user gradient on the sampled rows: dU_i = coef_i (sum_j p_ij I_j - I_{i + 3B})
    P = torch.softmax(S, dim=1)
    dU_ref = coef[rows].double()[:, None] * (P @ I_all.double() - I_all[rows + 3 * B].double())
    assert torch.allclose(dU.cpu()[rows].double(), dU_ref, atol=1e-9, rtol=2e-4)
    # sampled item rows: dI[j] = sum_i coef_i (p_ij - [j == i + 3B]) U_i needs every user -> fp64 on the GPU
    items = torch.tensor([0, 12345, 3 * B, 3 * B + 4097, 4 * B - 1, W * B - 1])
    Sd = U.to(DEV).double() @ I_all[items].to(DEV).double().t()
    lse_all = torch.logsumexp(U.to(DEV).double() @ I_all.to(DEV).double().t(), 1)
    G = torch.exp(Sd - lse_all[:, None])
    for k, j in enumerate(items.tolist()):
        if 3 * B <= j < 4 * B:
            G[j - 3 * B, k] -= 1.0
    ref = (G * coef.to(DEV).double()[:, None]).t() @ U.to(DEV).double()
    got = dI[items.to(DEV)].double()
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-12


def test_split_fp16_p_shape_whole_step_vs_oracle(T, monkeypatch):
    """EXPLORATORY path: one whole P-shape train step (B = 8192, D = 128, tables shrunk to what the CPU oracle sweeps in
    seconds) with TT_CE_F16X2 through the MODULE path against oracle/cpu_ref.py: loss 1e-4 (the north star's tolerance), the
    looked-up rows and the dense parameters at the fp32 path's criteria."""
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    ops, N = T
    monkeypatch.setattr(ops, "_CE_F16X2", True)
    n_users, n_items, D, F, B = 50_000, 100_000, 128, 8, 8192
    torch.manual_seed(0)
    model = A.TwoTowerBaseRetrieval(10, n_users, D, F, n_items, D, F, [1.0], A.BaselineMIPSModule(64, D))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("embedding_arch.weight") or n.endswith("tower_arch.weight"):
                p.mul_(0.5)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3)
    gen = torch.Generator().manual_seed(3)
    batch = (torch.randint(0, n_users, (B,), generator=gen), torch.randn(B, F, generator=gen), torch.randint(0, n_items, (B, 2), generator=gen),
             torch.randint(0, n_items, (B,), generator=gen), torch.randn(B, F, generator=gen), torch.randint(0, 10, (B,), generator=gen),
             torch.randint(0, 2, (B, 1), generator=gen).float())
    calls = []
    lib = N.load()
    real = lib.tt_ce16_bwd_recompute
    monkeypatch.setattr(lib, "tt_ce16_bwd_recompute", lambda *a: (calls.append(1), real(*a))[1])
    loss = model.train_forward(*[t.to(DEV) for t in batch])
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert calls == [1]
    state = R.AdamState(params)
    want = R.train_step(params, state, batch, torch.tensor([1.0]))
    assert abs(float(loss) - want) < 1e-4, (float(loss), want)
    for k, v in model.state_dict().items():
        err = (v.cpu() - params[k]).abs()
        noise_only = k in ("item_tower_arch.bias", "item_features_arch.2.bias")
        assert float(err.max()) <= 2.2e-3, (k, float(err.max()))
        if not noise_only:
            assert float((err > 5e-6).float().mean()) <= 2e-3, (k, float((err > 5e-6).float().mean()))


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


# ------------------------------------------------------------------ BASELINE configs at their stated size
def test_c2_whole_train_step_vs_oracle():
    """BASELINE config 2 end to end at its stated size (N_u = N_i = 1 M, D = 128, B = 4096, F = 8): two whole
    `train_forward -> zero_grad -> backward -> step` iterations (ref:train/train.py:112-125) on the HIP path vs
    oracle/cpu_ref.train_step on the same weights and batches: losses 1e-4, every dense parameter, 256 sampled
    looked-up and 256 sampled never-looked-up rows of each table.  The second step makes the rows of step 1
    move by momentum alone -- the zero-gradient sweep's arithmetic."""
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    NU = NI = 1_000_000
    D, F, B = 128, 8, 4096
    torch.manual_seed(0)
    mips = A.BaselineMIPSModule(corpus_size=64, embedding_dim=D)
    model = A.TwoTowerBaseRetrieval(10, NU, D, F, NI, D, F, [1.0], mips)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    g = torch.Generator().manual_seed(1234)
    batches = []
    for _ in range(2):
        batches.append([torch.randint(0, NU, (B,), generator=g), torch.randn(B, F, generator=g),
                        torch.randint(0, NI, (B, 4), generator=g), torch.randint(0, NI, (B,), generator=g),
                        torch.randn(B, F, generator=g), torch.randint(0, 10, (B,), generator=g),
                        torch.randint(0, 2, (B, 1), generator=g).float()])
    batches[1][0][:64] = batches[0][0][:64]  # users and items seen in BOTH steps
    batches[1][3][:64] = batches[0][3][:64]
    state = R.AdamState(params)
    got, want = [], []
    for b in batches:
        loss = model.train_forward(*[t.to(DEV) for t in b])
        opt.zero_grad()
        loss.backward()
        opt.step()
        got.append(loss.item())
        want.append(R.train_step(params, state, b, torch.tensor([1.0])))
    assert np.allclose(got, want, atol=1e-4), (got, want)
    sd = model.state_dict()
    for k, v in sd.items():
        if "embedding_arch" in k:
            continue
        # item_tower_arch.bias / item_features_arch.2.bias: analytically zero gradient
        noise_only = k in ("item_tower_arch.bias", "item_features_arch.2.bias")
        # Adam's first updates are lr * g / (|g| + eps): the few elements whose gradient happens to be ~1e-4 of the
        # typical size turn a 1e-7 relative summation-order difference into a ~1e-5 step difference (any two fp32
        # implementations do; tests/test_gpu_parallel.py).  So: all but <= 0.2 % of the elements within 5e-6, every
        # element within the steps * lr bound.
        err = (v.cpu() - params[k]).abs()
        assert float(err.max()) <= 2 * 2 * 1e-3 * 1.05, (k, float(err.max()))
        if not noise_only:
            assert float((err > 5e-6).float().mean()) <= 2e-3, (k, float((err > 5e-6).float().mean()), float(err.max()))
    pick = torch.Generator().manual_seed(5)
    for key, col, n_rows in (("user_id_embedding_arch.weight", 0, NU), ("item_id_embedding_arch.weight", 3, NI)):
        touched = torch.unique(torch.cat([b[col] for b in batches]))
        hit = touched[torch.randperm(touched.numel(), generator=pick)[:256]]
        hit = torch.cat([hit, batches[0][col][:64]])  # the rows both steps looked up
        mask = torch.ones(n_rows, dtype=torch.bool)
        mask[touched] = False
        cold = torch.nonzero(mask).flatten()
        cold = cold[torch.randperm(cold.numel(), generator=pick)[:256]]
        table = sd[key]
        assert torch.allclose(table[hit.to(DEV)].cpu(), params[key][hit], atol=5e-6), key
        assert torch.equal(table[cold.to(DEV)].cpu(), params[key][cold]), key
        # whole-table property: a row nobody looked up has m = v = 0 and must be bit-identical to its initial value
        st = opt.state[getattr(model, key.split(".")[0]).weight]
        assert float(st["exp_avg"][cold.to(DEV)].abs().max()) == 0.0


def _dense_params_close(sd, params, steps, noise_keys=("item_tower_arch.bias", "item_features_arch.2.bias")):
    """Dense parameters after `steps` Adam steps vs the oracle's: all but <= 0.2 % of a tensor's elements within 5e-6,
    every element within the steps * lr bound (Adam's first updates are lr * g / (|g| + eps): an element whose
    gradient is ~1e-4 of the typical size turns a 1e-7 summation-order difference into a 1e-5 step difference)."""
    for k, v in sd.items():
        if "embedding_arch" in k or k not in params:
            continue
        err = (v.cpu() - params[k]).abs()
        assert float(err.max()) <= 2 * steps * 1e-3 * 1.05, (k, float(err.max()))
        if k not in noise_keys:
            assert float((err > 5e-6).float().mean()) <= 2e-3, (k, float((err > 5e-6).float().mean()), float(err.max()))


def _sampled_rows_close(sd, opt, model, params, key, looked_up, n_rows, seed):
    pick = torch.Generator().manual_seed(seed)
    touched = torch.unique(looked_up)
    hit = touched[torch.randperm(touched.numel(), generator=pick)[:256]]
    mask = torch.ones(n_rows, dtype=torch.bool)
    mask[touched] = False
    cold = torch.nonzero(mask).flatten()
    cold = cold[torch.randperm(cold.numel(), generator=pick)[:256]]
    table = sd[key]
    assert torch.allclose(table[hit.to(DEV)].cpu(), params[key][hit], atol=5e-6), key
    assert torch.equal(table[cold.to(DEV)].cpu(), params[key][cold]), key
    st = opt.state[getattr(model, key.split(".")[0]).weight]
    assert float(st["exp_avg"][cold.to(DEV)].abs().max()) == 0.0
    return hit


def test_c3_whole_train_step_vs_oracle():
    """BASELINE config 3 end to end at its stated size (C2 + H = 50, 4 heads, 3 layers, B = 4096, 1 M items): one
    whole `train_forward -> zero_grad -> backward -> step` (ref:train/train.py:112-125 over
    ref:src/two_tower_with_user_history_encoder.py:85-122) on the HIP path vs oracle/cpu_ref.train_step: loss 1e-4,
    every dense parameter incl. the encoder's twelve, 256 sampled looked-up rows of each table -- the item rows
    include rows fed by BOTH an id lookup and history lookups -- and 256 never-looked-up rows (bit-identical)."""
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    NU = NI = 1_000_000
    D, F, B, H = 128, 8, 4096, 50
    torch.manual_seed(0)
    mips = A.BaselineMIPSModule(corpus_size=64, embedding_dim=D)
    model = A.TwoTowerWithUserHistoryEncoder(10, NU, D, F, H, NI, D, F, [1.0], mips)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    g = torch.Generator().manual_seed(4321)
    b = [torch.randint(0, NU, (B,), generator=g), torch.randn(B, F, generator=g),
         torch.randint(0, NI, (B, H), generator=g), torch.randint(0, NI, (B,), generator=g),
         torch.randn(B, F, generator=g), torch.randint(0, 10, (B,), generator=g),
         torch.randint(0, 2, (B, 1), generator=g).float()]
    b[2][:128, 0] = b[3][:128]      # items that are somebody's positive AND in histories (most recent slot ...
    b[2][128:256, 7] = b[3][:128]   # ... and a middle slot of other users)
    loss = model.train_forward(*[t.to(DEV) for t in b])
    opt.zero_grad()
    loss.backward()
    opt.step()
    state = R.AdamState(params)
    want = R.train_step(params, state, b, torch.tensor([1.0]), with_history=True, heads=4, pos_table=R.positional_table(H, D))
    assert abs(loss.item() - want) < 1e-4, (loss.item(), want)
    sd = model.state_dict()
    _dense_params_close(sd, params, 1)
    _sampled_rows_close(sd, opt, model, params, "user_id_embedding_arch.weight", b[0], NU, 5)
    _sampled_rows_close(sd, opt, model, params, "item_id_embedding_arch.weight", torch.cat([b[2].reshape(-1), b[3]]), NI, 6)
    both = b[3][:128]
    assert torch.allclose(sd["item_id_embedding_arch.weight"][both.to(DEV)].cpu(), params["item_id_embedding_arch.weight"][both],
                          atol=5e-6)


def _compact_oracle_step(model, b, D, with_hist=False):
    """The oracle's train step on a COMPACT copy of the problem: a step reads and moves only the looked-up rows, so
    the tables are cut down to those rows (ids remapped onto 0..n-1, same order statistics: duplicates stay duplicates)
    and oracle/cpu_ref.train_step runs on that.  Returns (loss, unique user ids, their rows after the step, unique item
    ids, their rows after the step, dense parameters after the step).  Lets a 100 M-row configuration be checked
    without 205 GB of host memory."""
    from oracle import cpu_ref as R
    sd = model.state_dict()
    uu, ui = torch.unique(b[0]), torch.unique(torch.cat([b[2].reshape(-1), b[3]]) if with_hist else b[3])
    params = {}
    for k, v in sd.items():
        if k == "user_id_embedding_arch.weight":
            params[k] = v[uu.to(v.device)].cpu()
        elif k == "item_id_embedding_arch.weight":
            params[k] = v[ui.to(v.device)].cpu()
        else:
            params[k] = v.detach().cpu().clone()
    bc = list(b)
    bc[0] = torch.searchsorted(uu, b[0])
    bc[3] = torch.searchsorted(ui, b[3])
    bc[2] = torch.searchsorted(ui, b[2]) if with_hist else torch.zeros_like(b[2])
    state = R.AdamState(params)
    loss = R.train_step(params, state, bc, torch.tensor([1.0]))
    return loss, uu, ui, params


def test_c4_100m_rows_one_step_sampled():
    """BASELINE config 4's table on ONE MI355X (N_i = 100 M rows x 128 = 51 GB, 154 GB with both Adam moments; each of
    the 8 ranks of the sharded plan holds an eighth): one whole train step (ref:train/train.py:112-125) at B = 8192
    through the module path vs the oracle on the compact problem -- loss 1e-4, 256 sampled looked-up rows of each
    table, the dense parameters, and never-looked-up rows on both sides of the 2^31- and 2^33-element lines
    bit-identical with zero moments."""
    import two_tower_models_amd as A
    free, _total = torch.cuda.mem_get_info()
    if free < 200e9:
        pytest.skip("needs ~170 GB of free HBM")
    NU, NI = 1_000_000, 100_000_000
    D, F, B = 128, 8, 8192
    torch.manual_seed(0)
    with torch.device(DEV):
        mips = A.BaselineMIPSModule(corpus_size=64, embedding_dim=D)
        model = A.TwoTowerBaseRetrieval(10, NU, D, F, NI, D, F, [1.0], mips)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    g = torch.Generator().manual_seed(777)
    b = [torch.randint(0, NU, (B,), generator=g), torch.randn(B, F, generator=g),
         torch.randint(0, NI, (B, 4), generator=g), torch.randint(0, NI, (B,), generator=g),
         torch.randn(B, F, generator=g), torch.randint(0, 10, (B,), generator=g),
         torch.randint(0, 2, (B, 1), generator=g).float()]
    b[3][:8] = torch.tensor([0, 1, NI - 1, NI - 2, (1 << 31) // D, (1 << 31) // D - 1, (1 << 33) // D, (1 << 33) // D + 1])
    b[3][8:16] = b[3][:8]  # duplicates across the lines
    want, uu, ui, params = _compact_oracle_step(model, b, D)
    itab = model.item_id_embedding_arch.weight
    cold = torch.tensor([(1 << 31) // D + 7, (1 << 31) // D - 9, (1 << 33) // D + 5, NI - 3, 2, 50_000_001, 87_654_321])
    cold = cold[~torch.isin(cold, b[3])]
    cold0 = itab.detach()[cold.to(DEV)].clone()
    loss = model.train_forward(*[t.to(DEV) for t in b])
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert abs(loss.item() - want) < 1e-4, (loss.item(), want)
    sd = model.state_dict()
    pick = torch.Generator().manual_seed(3)
    for key, uniq in (("user_id_embedding_arch.weight", uu), ("item_id_embedding_arch.weight", ui)):
        sel = torch.randperm(uniq.numel(), generator=pick)[:256]
        if key.startswith("item"):
            sel = torch.unique(torch.cat([sel, torch.searchsorted(uniq, b[3][:8])]))
        assert torch.allclose(sd[key][uniq[sel].to(DEV)].cpu(), params[key][sel], atol=5e-6), key
    assert torch.equal(itab.detach()[cold.to(DEV)], cold0)
    st = opt.state[itab]
    assert float(st["exp_avg"][cold.to(DEV)].abs().max()) == 0.0 and float(st["exp_avg_sq"][cold.to(DEV)].abs().max()) == 0.0
    assert bool((st["exp_avg"][b[3][:8].to(DEV)].abs().sum(1) > 0).all())
    _dense_params_close(sd, params, 1)


def test_c1_shape_train_steps_vs_oracle():
    """BASELINE config 1 at its stated shape (the reference's CPU-plumbing case: 1 K users x 10 K items, d = 32,
    B = 128, F = 8; ref:train/train.py defaults scaled as BASELINE.json states them): three whole train steps through the
    HIP path vs oracle/cpu_ref.train_step on the full tables -- losses 1e-4, both tables, every dense parameter."""
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    NU, NI, D, F, B = 1024, 10_000, 32, 8, 128
    torch.manual_seed(0)
    mips = A.BaselineMIPSModule(corpus_size=64, embedding_dim=D)
    model = A.TwoTowerBaseRetrieval(10, NU, D, F, NI, D, F, [1.0], mips)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    g = torch.Generator().manual_seed(99)
    state = R.AdamState(params)
    got, want = [], []
    for _ in range(3):
        b = [torch.randint(0, NU, (B,), generator=g), torch.randn(B, F, generator=g),
             torch.randint(0, NI, (B, 10), generator=g), torch.randint(0, NI, (B,), generator=g),
             torch.randn(B, F, generator=g), torch.randint(0, 10, (B,), generator=g),
             torch.randint(0, 2, (B, 1), generator=g).float()]
        loss = model.train_forward(*[t.to(DEV) for t in b])
        opt.zero_grad()
        loss.backward()
        opt.step()
        got.append(loss.item())
        want.append(R.train_step(params, state, b, torch.tensor([1.0])))
    assert np.allclose(got, want, atol=1e-4), (got, want)
    sd = model.state_dict()
    _dense_params_close(sd, params, 3)
    for key in ("user_id_embedding_arch.weight", "item_id_embedding_arch.weight"):
        err = (sd[key].cpu() - params[key]).abs()
        assert float(err.max()) <= 2 * 3 * 1e-3 * 1.05 and float((err > 5e-6).float().mean()) <= 2e-3, (key, float(err.max()))


def test_p_shape_train_step_vs_oracle():
    """The headline shape itself (BASELINE.json's metric: N_i = 10 M, N_u = 1 M, D = 128, B = 8192, F = 8): one whole
    step vs oracle/cpu_ref.train_step (about ten seconds of CPU), same checks as C2."""
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    NU, NI = 1_000_000, 10_000_000
    D, F, B = 128, 8, 8192
    torch.manual_seed(0)
    mips = A.BaselineMIPSModule(corpus_size=64, embedding_dim=D)
    model = A.TwoTowerBaseRetrieval(10, NU, D, F, NI, D, F, [1.0], mips)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    g = torch.Generator().manual_seed(99)
    b = [torch.randint(0, NU, (B,), generator=g), torch.randn(B, F, generator=g),
         torch.randint(0, NI, (B, 4), generator=g), torch.randint(0, NI, (B,), generator=g),
         torch.randn(B, F, generator=g), torch.randint(0, 10, (B,), generator=g),
         torch.randint(0, 2, (B, 1), generator=g).float()]
    b[3][:32] = b[3][32:64]  # duplicate positives inside the batch
    loss = model.train_forward(*[t.to(DEV) for t in b])
    opt.zero_grad()
    loss.backward()
    opt.step()
    state = R.AdamState(params)
    want = R.train_step(params, state, b, torch.tensor([1.0]))
    assert abs(loss.item() - want) < 1e-4, (loss.item(), want)
    sd = model.state_dict()
    _dense_params_close(sd, params, 1)
    _sampled_rows_close(sd, opt, model, params, "user_id_embedding_arch.weight", b[0], NU, 7)
    _sampled_rows_close(sd, opt, model, params, "item_id_embedding_arch.weight", b[3], NI, 8)
    dup = b[3][:32]
    assert torch.allclose(sd["item_id_embedding_arch.weight"][dup.to(DEV)].cpu(), params["item_id_embedding_arch.weight"][dup], atol=5e-6)


def _fp64_scores(q_eff, corpus, chunk=1_000_000):
    """[B, C] fp64 checker of q . corpus^T, computed on the device in row chunks (test infrastructure)."""
    out = torch.empty(q_eff.shape[0], corpus.shape[0], dtype=torch.float64, device=corpus.device)
    for lo in range(0, corpus.shape[0], chunk):
        out[:, lo:lo + chunk] = q_eff.double() @ corpus[lo:lo + chunk].double().t()
    return out


@pytest.mark.parametrize("bf16", [False, True])
def test_mips_10m_corpus_k1000_config5_size(T, bf16):
    """BASELINE config 5's corpus at its stated size: C = 10 M, D = 128, K = 1000 (fp32 corpus 5.12 GB, bf16
    2.56 GB -- both past 2^31 bytes, where 32-bit offsets would wrap).  Random corpus: every picked score against an
    fp64 checker, k-th-score optimality, order, range, no duplicates; rows planted in the LAST 1000 rows of the
    corpus must be found."""
    import two_tower_models_amd as A
    C_, D, B, K = 10_000_000, 128, 8, 1000
    g = torch.Generator(device=DEV).manual_seed(13)
    corpus = torch.randn(C_, D, device=DEV, generator=g)
    q = torch.randn(B, D, device=DEV, generator=g)
    corpus[C_ - 1000:] *= 3.0  # the tail of the corpus holds most of the winners
    m = A.BaselineMIPSModule(corpus_size=8, embedding_dim=D)
    m.corpus, m.corpus_size = corpus, C_
    if bf16:
        m.use_bf16_storage()
        del corpus
        q_eff = q.to(torch.bfloat16).float()
    else:
        q_eff = q
    idx, sc = m.search(q, K)
    assert idx.dtype == torch.int64 and idx.shape == (B, K) and bool(((idx >= 0) & (idx < C_)).all())
    assert bool((sc[:, 1:] <= sc[:, :-1]).all())
    assert all(len(set(r.tolist())) == K for r in idx.cpu())
    full = _fp64_scores(q_eff, m.corpus)
    picked = torch.gather(full, 1, idx)
    assert torch.allclose(picked, sc.double(), atol=5e-4)
    ref = torch.topk(full, K, dim=1)
    assert bool((picked.min(1).values >= ref.values[:, -1] - 5e-4).all())
    overlap = np.mean([len(set(a.tolist()) & set(b.tolist())) / K for a, b in zip(idx.cpu(), ref.indices.cpu())])
    assert overlap > 0.998
    assert float((idx >= C_ - 1000).float().mean()) > 0.05  # the planted tail rows (3x scale) were reached
    # the gathered rows of forward() (ref:src/baseline_mips_module.py:63-69) at offsets past 2^31 bytes
    idx2, sc2, emb = m(q[:2], 10)
    assert torch.equal(idx2, idx[:2, :10]) and torch.equal(emb, m.corpus[idx2].float())


@pytest.mark.parametrize("bf16,B", [(False, 8), (True, 8), (True, 72)])
def test_mips_10m_exact_arithmetic_corpus_bit_exact_order(T, bf16, B):
    """C = 10 M with small-integer embeddings (dot products exact in fp32 under any summation order, lossless in
    bf16): indices and scores must equal the CPU oracle's (score desc, index asc) order BIT FOR BIT, ties
    included (the index digits repeat every 65 536 rows, so equal scores are common).  B = 8: the shared-query
    form of pass 1 with 64-row groups; B = 72 (bf16): 256-row chunks / 128-row groups, the throughput form."""
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    C_, D, K = 10_000_000, 128, 1000
    g = torch.Generator(device=DEV).manual_seed(17)
    corpus = torch.randint(-1, 2, (C_, D), device=DEV, generator=g, dtype=torch.int8).float()
    i = torch.arange(C_, device=DEV)
    corpus[:, D - 3] = ((i & 63) - 32).float()
    corpus[:, D - 2] = (((i >> 6) & 63) - 32).float()
    corpus[:, D - 1] = (((i >> 12) & 15) - 8).float()
    q = torch.randint(-1, 2, (B, D), device=DEV, generator=g, dtype=torch.int8).float()
    q[:, D - 3], q[:, D - 2], q[:, D - 1] = 2.0 ** -6, 2.0 ** -12, 2.0 ** -16
    m = A.BaselineMIPSModule(corpus_size=8, embedding_dim=D)
    m.corpus, m.corpus_size = corpus, C_
    host = corpus.cpu()
    if bf16:
        m.use_bf16_storage()
        assert torch.equal(m.corpus[-4096:].float(), corpus[-4096:])  # lossless
    del corpus
    idx, sc = m.search(q, K)
    want_idx, want_sc, _ = R.mips_topk(q.cpu(), host, K, chunk=2)
    assert torch.equal(sc.cpu(), want_sc)
    assert torch.equal(idx.cpu(), want_idx)


@pytest.mark.parametrize("B", [8, 72, 300])
def test_mips_split_fp16_scoring_exact_arithmetic_and_random(T, B):
    """EXPLORATORY TT_F16X2 scoring (BaselineMIPSModule.use_split_fp16_scoring: fp32 corpus + its two-term fp16 split, three
    fp16 MFMA products per product).  (1) exact-arithmetic corpus of 3 M rows: indices AND scores bit for bit equal to the CPU
    oracle's (score desc, index asc) order, ties included -- the fp32 path's contract (B = 8: shared-query pass 1, 72 /
    300: the throughput forms); (2) random corpus: every picked score against a float64 checker at the fp32 path's
    tolerance, k-th-score optimality, overlap with the float64 top-K."""
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    C_, D, K = 3_000_000, 128, 500
    g = torch.Generator(device=DEV).manual_seed(19)
    corpus = torch.randint(-1, 2, (C_, D), device=DEV, generator=g, dtype=torch.int8).float()
    i = torch.arange(C_, device=DEV)
    corpus[:, D - 3] = ((i & 63) - 32).float()
    corpus[:, D - 2] = (((i >> 6) & 63) - 32).float()
    corpus[:, D - 1] = (((i >> 12) & 15) - 8).float()
    q = torch.randint(-1, 2, (B, D), device=DEV, generator=g, dtype=torch.int8).float()
    q[:, D - 3], q[:, D - 2], q[:, D - 1] = 2.0 ** -6, 2.0 ** -12, 2.0 ** -16
    m = A.BaselineMIPSModule(corpus_size=8, embedding_dim=D)
    m.corpus, m.corpus_size = corpus, C_
    m.use_split_fp16_scoring()
    idx, sc = m.search(q, K)
    nq = min(B, 16)  # the oracle is a CPU matmul: a sample of the queries
    want_idx, want_sc, _ = R.mips_topk(q[:nq].cpu(), corpus.cpu(), K, chunk=2)
    assert torch.equal(sc[:nq].cpu(), want_sc)
    assert torch.equal(idx[:nq].cpu(), want_idx)
    idx32, sc32 = A.BaselineMIPSModule.search(m.use_split_fp16_scoring(False), q, K)  # and the fp32-MFMA path agrees
    assert torch.equal(idx32, idx) and torch.equal(sc32, sc)
    # random data: the corpus is REPLACED and then refilled in place -- the split must follow it both times
    corpus = torch.randn(C_, D, device=DEV, generator=g)
    q = torch.randn(B, D, device=DEV, generator=g)
    m.use_split_fp16_scoring()
    m.search(q[:4], 5)
    m.corpus = torch.zeros_like(corpus)
    m.search(q[:4], 5)
    m.corpus.copy_(corpus)
    idx, sc = m.search(q, K)
    assert bool((sc[:, 1:] <= sc[:, :-1]).all()) and bool(((idx >= 0) & (idx < C_)).all())
    assert all(len(set(r.tolist())) == K for r in idx[:8].cpu())
    full = _fp64_scores(q[:32], corpus)
    picked = torch.gather(full, 1, idx[:32])
    assert torch.allclose(picked, sc[:32].double(), atol=5e-5)  # (the fp32 path's test allows 5e-4)
    ref = torch.topk(full, K, dim=1)
    assert bool((picked.min(1).values >= ref.values[:, -1] - 5e-5).all())
    overlap = np.mean([len(set(a.tolist()) & set(b.tolist())) / K for a, b in zip(idx[:32].cpu(), ref.indices.cpu())])
    assert overlap > 0.999


def test_table_beyond_2_31_elements_gather_and_adam(T):
    """A table with more than 2^31 ELEMENTS (20 M x 128 = 2.56 G floats, 10.2 GB; 30.7 GB with both moments -- what
    one rank of BASELINE config 4 holds at N = 8 is 12.5 M rows): gather, the row plan and two dense-exact Adam
    steps with rows on both sides of the 2^31-element line, sampled rows vs oracle/cpu_ref.adam_update."""
    from oracle import cpu_ref as R
    ops, N = T
    lib = N.load()
    n_rows, D, n = 20_000_000, 128, 8192
    line = (1 << 31) // D  # first row whose elements sit past 2^31
    g = torch.Generator(device=DEV).manual_seed(23)
    Wd = torch.randn(n_rows, D, device=DEV, generator=g)
    Md, Vd = torch.zeros_like(Wd), torch.zeros_like(Wd)
    cg = torch.Generator().manual_seed(29)
    steps = []
    for _ in (1, 2):
        ids = torch.cat([torch.randint(0, n_rows, (n - 2048,), generator=cg),
                         torch.randint(line, n_rows, (2040,), generator=cg),
                         torch.tensor([n_rows - 1, n_rows - 2, line, line - 1, line + 1, 0, 1, n_rows - 1])])
        steps.append((ids, torch.randn(n, D, generator=cg) * 0.01))
    steps[1][0][:32] = steps[0][0][:32]
    all_ids = torch.cat([s[0] for s in steps])
    # gather across the line
    out = torch.empty(all_ids.numel(), D, device=DEV)
    ops.gather_rows_into(Wd, all_ids.to(DEV), out)
    assert torch.equal(out, Wd[all_ids.to(DEV)])
    sample = torch.unique(torch.cat([steps[0][0][:150], steps[0][0][-160:], steps[1][0][-160:]]))
    W0 = Wd[sample.to(DEV)].cpu()
    cold = torch.tensor([line - 2, line + 2, n_rows - 3, 12_345_678, 19_999_000])
    cold = cold[~torch.isin(cold, all_ids)]
    cold0 = Wd[cold.to(DEV)].clone()
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0, 0, 0, 0], dtype=torch.float64, device=DEV)
    for ids, rows in steps:
        N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "adv")
        plan = ops.RowPlan.from_grads([ops.RowGrad(ids.to(DEV), rows.to(DEV))], n_rows)
        wsp, wsn = ops._ws(torch.device(DEV), lib.tt_adam_table_workspace_bytes(plan.n, D), "adam_side")
        N.check(lib.tt_adam_table(Wd.data_ptr(), Md.data_ptr(), Vd.data_ptr(), n_rows, D, hyper.data_ptr(),
                                  C.byref(plan.sources), plan.n, plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                  plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), wsp, wsn, N.stream()), "adam")
    got = Wd[sample.to(DEV)].cpu()
    for j, k in enumerate(sample.tolist()):
        p, m_, v_ = W0[j].clone(), torch.zeros(D), torch.zeros(D)
        for step, (ids, rows) in enumerate(steps, start=1):
            hit = ids == k
            R.adam_update(p, rows[hit].sum(0) if bool(hit.any()) else torch.zeros(D), m_, v_, step)
        assert torch.allclose(got[j], p, atol=3e-6), k
    assert torch.equal(Wd[cold.to(DEV)], cold0)  # never looked up, m = v = 0: bit-identical
    # the moments past the line are non-zero exactly where rows were looked up
    tail_touched = torch.unique(all_ids[all_ids >= line])
    assert bool((Md[tail_touched.to(DEV)].abs().sum(1) > 0).all())
    assert float(Md[line:].abs().sum(1).gt(0).sum()) == float(tail_touched.numel())
