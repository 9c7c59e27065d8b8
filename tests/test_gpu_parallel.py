"""Row-sharded training THROUGH THE MODULE CLASSES (two_tower_models_amd.parallel) with the product kernels on one
MI355X: W ranks construct `TwoTowerBaseRetrieval` / `TwoTowerWithUserHistoryEncoder` / `TwoTowerWithDebiasing` under a
process group, run the reference loop (train_forward -> zero_grad -> backward -> DenseExactAdam.step,
ref:train/train.py:112-125) on their own B rows, and must reproduce `oracle.cpu_ref.train_step` on the CONCATENATED
batch of W*B rows (SURVEY.md 8e "parity definition"): loss 1e-4, every table row and every dense parameter.

RCCL refuses two ranks on the same device, so on the 1-GPU test box the ranks are separate processes that share cuda:0
and exchange through gloo (collectives.py stages device tensors through the host for gloo).  Everything except the
transport is the product path.  The nccl cases run the same workers one rank per device and skip on a 1-GPU box."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS = 3
UVW = 0.7

CASES = {
    # name: (model, cfg)
    "base_d128": ("base", dict(n_users=300, n_items=500, D=128, F=8, B=128, H=2)),
    # generic kernels (D = 40: no fused tower), tables that do not divide by the world size
    "base_ragged": ("base", dict(n_users=53, n_items=71, D=40, F=20, B=24, H=2)),
    # a batch larger than either table (every row looked up several times, by several ranks), fused-tower width 64
    "base_crowded": ("base", dict(n_users=37, n_items=29, D=64, F=5, B=96, H=3)),
    # TwoTowerWithUserHistoryEncoder: 4 heads x dh 32 (MFMA attention), B*H = 240 history rows per rank
    "hist": ("hist", dict(n_users=211, n_items=307, D=128, F=8, B=40, H=6)),
    # the BASELINE history length (attention tiles padded to 64) and a table smaller than one rank's history list
    "hist50": ("hist", dict(n_users=90, n_items=131, D=128, F=8, B=24, H=50)),
    # BASELINE's history length at a real per-rank batch, ids drawn from 40 items: every item row is looked up ~320 times per
    # rank and step (B*H = 12 800 history lookups + 256 item ids per rank), by all ranks -- the routed exchange at its most
    # duplicated (SURVEY section 7 "History all-to-all volume")
    "hist50_dup": ("hist", dict(n_users=90, n_items=40, D=128, F=8, B=256, H=50)),
    # the MARKED sweep on row blocks (optim._SPLIT_MIN_IDS lowered from 65 536: the owners mark the rows they serve instead
    # of parking them; the padding slots' sentinel ids and other ranks' rows are ignored by the marking)
    "hist_marked": ("hist", dict(n_users=211, n_items=307, D=128, F=8, B=40, H=6, mark_from=1)),
    "hist50_dup_marked": ("hist", dict(n_users=90, n_items=40, D=64, F=8, B=256, H=50, mark_from=1)),
    "base_marked": ("base", dict(n_users=300, n_items=500, D=128, F=8, B=64, H=2, mark_from=1)),
    # an item table with fewer rows than ranks x rows-per-rank: the last rank owns NO item row
    "hist_empty_block": ("hist", dict(n_users=338, n_items=9, D=64, F=20, B=33, H=1)),
    # BASELINE config 5's model in training: the fused debias head on the gathered batch
    "debias": ("debias", dict(n_users=211, n_items=307, D=128, F=8, B=40, H=6)),
    "debias_small": ("debias", dict(n_users=53, n_items=71, D=32, F=8, B=16, H=3)),
    # 1-D [B] labels (ref:train/train.py:53-55): the plain-mean quirk, group-wide
    "base_labels1d": ("base", dict(n_users=300, n_items=500, D=128, F=8, B=64, H=2, labels1d=True)),
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _case(name):
    """A named case, or any configuration as "json:[kind, {...}]" (tools/fuzz_sharded.py; the spawned ranks re-import this
    module, so the configuration has to travel in the name)."""
    import json
    if name.startswith("json:"):
        kind, cfg = json.loads(name[5:])
        return kind, cfg
    return CASES[name]


def build_case(case):
    """The WHOLE model on the CPU, seeded: every rank and the checking process construct the same parameters.  Scaled
    so that logits and attention scores are O(1) (with O(100) logits most rows saturate, p - 1 cancels catastrophically
    and whole rows carry a 1e-2 relative gradient error in ANY fp32 implementation)."""
    import two_tower_models_amd as A
    kind, cfg = _case(case)
    torch.manual_seed(0)
    mips = A.BaselineMIPSModule(corpus_size=16, embedding_dim=cfg["D"])
    kw = dict(num_items=5, user_id_hash_size=cfg["n_users"], user_id_embedding_dim=cfg["D"], user_features_size=cfg["F"],
              item_id_hash_size=cfg["n_items"], item_id_embedding_dim=cfg["D"], item_features_size=cfg["F"],
              user_value_weights=[UVW], mips_module=mips)
    if kind == "base":
        model = A.TwoTowerBaseRetrieval(**kw)
    elif kind == "hist":
        model = A.TwoTowerWithUserHistoryEncoder(user_history_seqlen=cfg["H"], **kw)
    else:
        model = A.TwoTowerWithDebiasing(user_history_seqlen=cfg["H"], **kw)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("embedding_arch.weight"):
                p.mul_(0.5)
            elif name.endswith("tower_arch.weight"):
                p.mul_(0.5 if kind == "base" else 0.3)
            elif "in_proj_weight" in name or "out_proj.weight" in name:
                p.mul_(0.5)
            elif name.startswith("position_bias_net_user_value"):
                p.mul_(0.2).add_(0.6)  # priors around the labels' scale: the clamps are exercised on a few rows only
            elif name.startswith("user_debias_net_user_value") and name.endswith("bias"):
                p.add_(0.7)
    return model


def make_batches(case, rank, n, seed=99):
    _, cfg = _case(case)
    gen = torch.Generator().manual_seed(seed + 1000 * rank)
    B, F = cfg["B"], cfg["F"]
    out = []
    for _ in range(n):
        labels = torch.randint(0, 2, (B,) if cfg.get("labels1d") else (B, 1), generator=gen).float()
        out.append((torch.randint(0, cfg["n_users"], (B,), generator=gen), torch.randn(B, F, generator=gen),
                    torch.randint(0, cfg["n_items"], (B, cfg["H"]), generator=gen),
                    torch.randint(0, cfg["n_items"], (B,), generator=gen), torch.randn(B, F, generator=gen),
                    torch.randint(0, 10, (B,), generator=gen), labels))
    return out


def init_pg(backend, rank, world, port):
    """gloo: every rank on cuda:0 (1-GPU test box).  nccl (= RCCL): one rank per device."""
    import torch.distributed as dist
    dev = torch.device(f"cuda:{rank}" if backend == "nccl" else "cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    return dev


def perturb_host_timing(seed, max_us=300):
    """Random 0..max_us sleeps after EVERY library call that enqueues work (monkeypatched _native.check): the host then runs
    sometimes ahead of the GPU, sometimes behind it, launch by launch.  Results must not depend on it (VERDICT r5 weak #6:
    three streams + engine callbacks + held sweeps once produced a schedule that depended on host timing)."""
    import random
    import time
    from two_tower_models_amd import _native as N
    rng, real = random.Random(seed), N.check

    def check(rc, what):
        real(rc, what)
        time.sleep(rng.random() * max_us * 1e-6)

    N.check = check
    return lambda: setattr(N, "check", real)


def _worker(rank, world, port, outdir, case, backend, transport, sharded_init, perturb_seed=None, steps=STEPS):
    _paths()
    import torch.distributed as dist
    import two_tower_models_amd as A
    from two_tower_models_amd import collectives, parallel
    dev = init_pg(backend, rank, world, port)
    restore = perturb_host_timing(perturb_seed + rank) if perturb_seed is not None else (lambda: None)
    mark_from = _case(case)[1].get("mark_from")
    if mark_from is not None:
        from two_tower_models_amd import optim
        optim._SPLIT_MIN_IDS = mark_from
    try:
        if transport == "native":
            from two_tower_models_amd.comm import NativeComm
            collectives.use_native_transport(NativeComm.from_torch_distributed(dev))
        whole = build_case(case)
        if sharded_init:
            # tables born sharded (`with parallel.row_sharded()`): the blocks are then FILLED from the whole model's rows so
            # that the run is comparable with the oracle; what is exercised is that construction path
            with parallel.row_sharded():
                model = build_case(case)
            with torch.no_grad():
                for (_, p), (_, q) in zip(model.named_parameters(), whole.named_parameters()):
                    sh = parallel.shard_of(p)
                    if sh is None:
                        p.copy_(q)
                    else:
                        assert p.shape[0] == max(sh.n_local, 1) and q.shape[0] == sh.n_rows
                        p[: sh.n_local].copy_(q[sh.lo:sh.hi])
        else:
            model = whole
        model = model.to(dev)
        parallel.shard_model_(model)
        assert parallel.is_sharded(model)
        opt = A.DenseExactAdam(model.parameters(), lr=1e-3)
        assert opt.overlap_sweep == "forward"
        batches = [tuple(t.to(dev) for t in b) for b in make_batches(case, rank, steps)]
        losses = []
        for i, b in enumerate(batches):
            loss = model.train_forward(*b)
            assert mark_from is None or all(ts.marked for ts in opt._begun.values())
            if i % 3 == 0 and i + 1 < len(batches):  # batch 1 is announced (routes planned one step ahead); batch 2 arrives unannounced
                parallel.plan_ahead(model._lookup_plan(batches[i + 1][0], batches[i + 1][2], batches[i + 1][3]))
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        torch.cuda.synchronize()
        sd = parallel.full_state_dict(model)
        shards = {n: (parallel.shard_of(p).lo, parallel.shard_of(p).hi) for n, p in model.named_parameters()
                  if parallel.shard_of(p) is not None}
        torch.save({"losses": losses, "sd": {k: v.cpu() for k, v in sd.items()}, "shards": shards,
                    "comm": dict(parallel.comm_bytes), "batches": [tuple(t.cpu() for t in b) for b in batches]},
                   os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        restore()
        collectives.use_native_transport(None)
        dist.destroy_process_group()


def resolve_world(world, backend):
    """nccl cases need one device per rank: skipped on the 1-GPU test boxes, run on any multi-GPU node
    ("all" = every device of the node)."""
    if backend != "nccl":
        return world
    n = torch.cuda.device_count()
    if world == "all":
        world = n
    if n < 2 or world > n:
        pytest.skip(f"RCCL case needs {world} devices, this box has {n}")
    return world


def oracle_run(case, res, world):
    from oracle import cpu_ref as R
    kind, cfg = _case(case)
    params = {k: v.detach().clone() for k, v in build_case(case).state_dict().items()}
    state = R.AdamState(params)
    kw = {}
    if kind != "base":
        kw = dict(with_history=True, heads=4, pos_table=R.positional_table(cfg["H"], cfg["D"]))
    if kind == "debias":
        kw["debias"] = R.debias_combined
    want = []
    for s in range(STEPS):
        cat = [torch.cat([res[r]["batches"][s][k] for r in range(world)]) for k in range(7)]
        want.append(R.train_step(params, state, cat, torch.tensor([UVW]), **kw))
    return want, params


def check_against_oracle(case, res, world, outlier_frac=2e-3):
    want, params = oracle_run(case, res, world)
    bad = []
    for r in range(world):
        assert np.allclose(res[r]["losses"], want, atol=1e-4), (res[r]["losses"], want)
        # Adam's early updates are lr * g / (|g| + eps)-like: the few elements whose gradient happens to be ~1e-4 of the
        # typical size turn a 1e-7 relative summation-order difference into a ~1e-5 step difference (the CPU restatement
        # run as 2 ranks shows the same outliers against the 1-rank reference).  So: all but <= 0.2 % of the elements
        # tight, every element within the steps * lr bound.
        for name, got in res[r]["sd"].items():
            ref = params[name]
            assert got.shape == ref.shape, name
            err = (got - ref).abs()
            if err.numel() == 0:
                continue
            assert float(err.max()) <= 2.2e-3 * STEPS, (name, r, float(err.max()))
            # zero true gradient, noise only: the item-side biases and the key third of every in_proj_bias
            noise_only = name in ("item_tower_arch.bias", "item_features_arch.2.bias") or name.endswith("in_proj_bias")
            if not noise_only and int((err > 5e-6).sum()) > max(2, int(outlier_frac * err.numel())):
                rows = (err.reshape(err.shape[0], -1) > 5e-6).any(1).nonzero().flatten()  # first and last offending row
                bad.append((name, r, int((err > 5e-6).sum()), err.numel(), float(err.max()), (int(rows[0]), int(rows[-1]))))
        # replicas stay bit-identical, and every rank assembled the same whole tables
        assert all(torch.equal(v, res[0]["sd"][k]) for k, v in res[r]["sd"].items())
    assert not bad, bad


@pytest.mark.parametrize("world,case,backend,transport,sharded_init", [
    (2, "base_d128", "gloo", "torch", False), (3, "base_ragged", "gloo", "torch", False),
    (3, "base_crowded", "gloo", "torch", True), (2, "base_labels1d", "gloo", "torch", False),
    (2, "hist", "gloo", "torch", False), (3, "hist50", "gloo", "torch", True), (4, "hist_empty_block", "gloo", "torch", False),
    (3, "hist50_dup", "gloo", "torch", False),
    (2, "hist_marked", "gloo", "torch", False), (3, "hist50_dup_marked", "gloo", "torch", True), (2, "base_marked", "gloo", "torch", False),
    (2, "debias", "gloo", "torch", False), (3, "debias_small", "gloo", "torch", True),
    # RCCL, one device per rank: skipped on the 1-GPU test boxes, run on any multi-GPU node
    (2, "base_d128", "nccl", "torch", False), ("all", "base_d128", "nccl", "torch", True),
    ("all", "base_ragged", "nccl", "torch", False), (2, "hist", "nccl", "torch", False), ("all", "debias", "nccl", "torch", False),
    # the C ABI's own collectives (tt_comm_*) instead of torch's process group
    (2, "base_d128", "nccl", "native", False), ("all", "hist", "nccl", "native", False)])
def test_sharded_modules_equal_reference_on_concatenated_batch(world, case, backend, transport, sharded_init, outlier_frac=2e-3):
    import torch.multiprocessing as mp
    _paths()
    world = resolve_world(world, backend)
    outdir = tempfile.mkdtemp()
    mp.spawn(_worker, args=(world, _free_port(), outdir, case, backend, transport, sharded_init), nprocs=world, join=True)
    res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    check_against_oracle(case, res, world, outlier_frac)
    assert "lookup_rows_alltoall" in res[0]["comm"] and "dense_grad_allreduce" in res[0]["comm"]
    # the shards tile the tables exactly
    _, cfg = _case(case)
    for name, n in (("user_id_embedding_arch.weight", cfg["n_users"]), ("item_id_embedding_arch.weight", cfg["n_items"])):
        assert sum(r["shards"][name][1] - r["shards"][name][0] for r in res) == n


# ------------------------------------------------------------------ sharded MIPS (BASELINE config 5) behind the drop-in API
def _serve_corpus(C, D, skewed=False):
    """fixture_gen's exact-arithmetic corpus; `skewed`: the first third of the rows doubled (exact) -- nearly every query's
    whole top-K then lies in block 0, which the first-try k' of parallel.sharded_topk has to detect."""
    import fixture_gen as fg
    corpus = torch.from_numpy(fg.exact_mips_corpus(C, D))
    if skewed:
        corpus[: C // 3] *= 2.0
    return corpus


def _serve_model(C, D, n_users, how, dev, skewed=False):
    """TwoTowerWithDebiasing (BASELINE config 5's model) whose user tower is the identity on the id embedding, so that
    forward()'s query embeddings are EXACTLY the user table's rows (x * 1 + 0 * anything): with the exact-arithmetic
    corpus / queries of fixture_gen the top-K then has one right answer, bit for bit."""
    import contextlib
    import fixture_gen as fg
    import two_tower_models_amd as A
    from two_tower_models_amd import parallel
    torch.manual_seed(0)
    ctx = parallel.row_sharded() if how == "born" else contextlib.nullcontext()
    with ctx:
        mips = A.BaselineMIPSModule(corpus_size=C, embedding_dim=D)
        model = A.TwoTowerWithDebiasing(num_items=1, user_id_hash_size=n_users, user_id_embedding_dim=D, user_features_size=8,
                                        user_history_seqlen=4, item_id_hash_size=50, item_id_embedding_dim=D, item_features_size=8,
                                        user_value_weights=[1.0], mips_module=mips)
    model = model.to(dev)
    if how == "cut":  # every rank holds the whole corpus; shard_model_ keeps its block
        model.mips_module.corpus = _serve_corpus(C, D, skewed).to(dev)
    parallel.shard_model_(model)
    queries = torch.from_numpy(fg.exact_mips_queries(n_users, D))
    sd = parallel.full_state_dict(model)
    sd["user_id_embedding_arch.weight"] = queries
    w = torch.zeros_like(sd["user_tower_arch.weight"])
    w[:, :D] = torch.eye(D)
    sd["user_tower_arch.weight"], sd["user_tower_arch.bias"] = w, torch.zeros(D)
    parallel.load_full_state_dict(model, sd)
    return model, queries


def _mips_worker(rank, world, port, outdir, C, K, D, bf16, how, backend="gloo", skewed=False):
    _paths()
    import torch.distributed as dist
    import fixture_gen as fg
    from two_tower_models_amd import parallel
    dev = init_pg(backend, rank, world, port)
    try:
        n_users, B = 64, 6
        model, _ = _serve_model(C, D, n_users, how, dev, skewed)
        m = model.mips_module
        corpus = _serve_corpus(C, D, skewed)
        _, lo, hi = parallel.block_range(C, rank, world)
        if how == "born":
            assert m.is_sharded() and m.corpus.shape[0] == hi - lo
            m.set_corpus(corpus[lo:hi].to(dev), bf16=bf16)
        elif bf16:
            m.use_bf16_storage()
        assert m.is_sharded() and m.corpus_size == C and m.corpus.shape[0] == hi - lo and m.corpus.is_cuda
        assert m.corpus.dtype == (torch.bfloat16 if bf16 else torch.float32)
        model.num_items = K
        g = torch.Generator().manual_seed(100 + rank)
        uid = torch.randint(0, n_users, (B,), generator=g)
        batch = (uid.to(dev), torch.randn(B, 8, generator=g).to(dev), torch.randint(0, 50, (B, 4), generator=g).to(dev))
        top = model(*batch)  # TwoTowerWithDebiasing.forward (ref:src/two_tower_base_retrieval.py:221-249), sharded; no optimiser, no no_grad
        with torch.no_grad():
            q = model.compute_user_embedding(*batch)
            idx, sc, emb = m(query_embedding=q, num_items=K)  # the reference's call, 3-tuple
        torch.save({"uid": uid, "top": top.cpu(), "q": q.cpu(), "idx": idx.cpu(), "sc": sc.cpu(), "emb": emb.cpu(),
                    "comm": dict(parallel.comm_bytes)}, os.path.join(outdir, f"mips{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,C,K,D,bf16,how,backend", [
    (1, 5000, 100, 128, False, "cut", "nccl1"), (2, 9000, 100, 128, False, "born", "gloo"),
    (2, 9000, 100, 128, True, "cut", "gloo"),
    (3, 200, 80, 64, True, "born", "gloo"),  # K larger than a block (67 rows), generic-width towers
    (4, 5, 3, 128, False, "cut", "gloo"),  # the last rank's corpus block is empty
    # K large enough for a first-try k' < K (parallel.first_try_k): enough on the plain corpus, NOT enough on the skewed one
    (4, 9000, 400, 128, True, "cut", "gloo"), (3, 9000, 300, 128, False, "born", "gloo-skewed"),
    (2, 9000, 100, 128, True, "born", "nccl"), ("all", 9000, 100, 128, False, "cut", "nccl")])
def test_sharded_model_forward_topk_bit_exact(world, C, K, D, bf16, how, backend):
    """Row N2: `TwoTowerWithDebiasing.forward()` on a row-sharded model (tables AND the MIPS corpus in row blocks) returns
    oracle.cpu_ref.mips_topk's indices over the WHOLE corpus bit for bit; BaselineMIPSModule.forward's 3-tuple: global
    int64 indices, bit-exact scores, embeddings == corpus[idx] fetched from their owners."""
    import torch.multiprocessing as mp
    _paths()
    import fixture_gen as fg
    from oracle import cpu_ref as R
    skewed = backend.endswith("-skewed")
    backend = backend.replace("-skewed", "")
    if backend == "nccl1":  # an RCCL group of one on the 1-GPU box
        backend = "nccl"
    else:
        world = resolve_world(world, backend)
    outdir = tempfile.mkdtemp()
    mp.spawn(_mips_worker, args=(world, _free_port(), outdir, C, K, D, bf16, how, backend, skewed), nprocs=world, join=True)
    corpus = _serve_corpus(C, D, skewed)
    queries = torch.from_numpy(fg.exact_mips_queries(64, D))
    for r in range(world):
        got = torch.load(os.path.join(outdir, f"mips{r}.pt"))
        assert torch.equal(got["q"], queries[got["uid"]])  # the identity tower: queries are the table rows, exactly
        want_idx, want_sc, want_emb = R.mips_topk(queries[got["uid"]], corpus, K)
        assert got["top"].dtype == torch.int64 and torch.equal(got["top"], want_idx)
        assert torch.equal(got["idx"], want_idx) and torch.equal(got["sc"], want_sc)
        assert got["emb"].dtype == torch.float32 and torch.equal(got["emb"], want_emb)
        if world > 1:
            assert {"mips_queries_allgather", "mips_lists_alltoall", "mips_rows_alltoall"} <= set(got["comm"])
            from two_tower_models_amd import parallel
            k1 = parallel.first_try_k(K, world)
            rounds = [k1, K] if (skewed and k1 < K) else [k1]
            assert got["comm"]["mips_lists_alltoall"] == (world - 1) * 6 * 12 * sum(rounds), (got["comm"], k1)


def test_mips_merge_kernel_on_hand_made_shard_lists():
    """tt_mips_merge: ties across shards order by index, -1 = "no candidate" padding."""
    from two_tower_models_amd import ops
    dev = torch.device("cuda:0")
    sc = torch.tensor([[5., 3., 3., 1., 5., 4., 3., 0., 9., 3., 2., 0.],
                       [1., 1., 1., 1., 1., 1., 1., 0., 1., 1., 0., 0.]])
    ix = torch.tensor([[10, 11, 12, 13, 20, 21, 22, -1, 3, 30, 31, -1],
                       [7, 8, 9, 10, 1, 2, 3, -1, 4, 5, -1, -1]])
    oi, os_ = ops.mips_merge(sc.to(dev), ix.to(dev), 5)
    assert oi.cpu().tolist() == [[3, 10, 20, 21, 11], [1, 2, 3, 4, 5]]
    assert os_.cpu().tolist() == [[9., 5., 5., 4., 3.], [1., 1., 1., 1., 1.]]


# ------------------------------------------------------------------ checkpoints and corpus serving, sharded
def _ckpt_worker(rank, world, port, outdir, backend):
    _paths()
    import torch.distributed as dist
    import two_tower_models_amd as A
    from two_tower_models_amd import parallel
    g = np.load(os.path.join(HERE, "golden", "g2_base_aligned.npz"))
    n_users, du, iu, n_items, di, ii, Tn, B, H = (int(v) for v in g["cfg"])
    uvw = [float(v) for v in g["uvw"]]
    dev = init_pg(backend, rank, world, port)
    try:
        Bl = B // world
        with parallel.row_sharded():
            model = A.TwoTowerBaseRetrieval(10, n_users, du, iu, n_items, di, ii, uvw,
                                            A.BaselineMIPSModule(corpus_size=n_items, embedding_dim=di)).to(dev)
        parallel.shard_model_(model)
        # a REFERENCE-format checkpoint (the parameters the reference model was created with) into the shards ...
        parallel.load_full_state_dict(model, {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p.")})
        opt = A.DenseExactAdam(model.parameters(), lr=1e-3)
        names = ("user_id", "user_features", "user_history", "item_id", "item_features", "position", "labels")
        losses = []
        for s in range(3):  # ... the reference's three batches, split by rank ...
            b = [torch.from_numpy(g[f"step{s}.in.{n}"])[rank * Bl:(rank + 1) * Bl].to(dev) for n in names]
            loss = model.train_forward(*b)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        sd = parallel.full_state_dict(model)  # ... and back out under the reference's Parameter names
        # serve the trained item table: this rank's catalogue block through the item tower -> the model's own (row-sharded)
        # mips_module, i.e. model.index_corpus(...) + model.forward(...)
        feats = torch.from_numpy(g["step0.in.item_features"])  # any [*, II] features: row r of the catalogue gets row r % B
        cat_feats = feats[torch.arange(n_items) % B]
        sh = parallel.shard_of(model.item_id_embedding_arch.weight)
        mips = parallel.index_corpus_sharded(model, cat_feats[sh.lo:sh.hi])
        assert mips is model.mips_module and mips.is_sharded() and mips.corpus_size == n_items
        block_local = mips.corpus.clone()
        # ... and the general form: any ids (here the same catalogue, permuted across the ranks' blocks -> routed lookups)
        perm = torch.arange(n_items).flip(0)
        model.index_corpus(perm[sh.lo:sh.hi].to(dev), cat_feats[perm][sh.lo:sh.hi].to(dev), chunk=97)
        block_routed = model.mips_module.corpus.clone()
        parallel.index_corpus_sharded(model, cat_feats[sh.lo:sh.hi])
        torch.save({"losses": losses, "sd": {k: v.cpu() for k, v in sd.items()}}, os.path.join(outdir, f"ckpt{rank}.pt"))
        dist.barrier()
        # queries: the user embeddings of the trained model, from a single-device module fed the gathered state
        single = A.TwoTowerBaseRetrieval(10, n_users, du, iu, n_items, di, ii, uvw,
                                         A.BaselineMIPSModule(corpus_size=n_items, embedding_dim=di))
        single.load_state_dict(sd)
        single = single.to(dev)
        with torch.no_grad():
            single.index_corpus(torch.arange(n_items, device=dev), cat_feats.to(dev))
            users = [torch.from_numpy(g[f"step2.in.{n}"])[rank * Bl:(rank + 1) * Bl].to(dev) for n in names[:3]]
            want_top = single(*users)
            q = single.compute_user_embedding(*users)
            # the sharded model's own inference path (un-announced routed lookups) gives the same query embeddings
            q_sharded = model.compute_user_embedding(*users)
            top_sharded = model(*users)  # forward() of the SHARDED model: routed lookups + the sharded corpus, no other call
        assert torch.allclose(q_sharded, q, atol=1e-6)
        idx, _ = mips.search(q, 10)
        # the routed index_corpus saw the catalogue reversed: its block r holds the embeddings of items perm[lo:hi]
        whole = single.mips_module.corpus
        assert torch.allclose(block_local, whole[sh.lo:sh.hi], atol=1e-6)
        assert torch.allclose(block_routed, whole[perm[sh.lo:sh.hi].to(dev)], atol=1e-6)
        torch.save({"idx": idx.cpu(), "want": want_top.cpu(), "top_sharded": top_sharded.cpu()},
                   os.path.join(outdir, f"serve{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo"), (2, "nccl"), ("all", "nccl")])
def test_sharded_checkpoint_adaptor_and_corpus_serving(world, backend):
    """SURVEY 8f item 4 through the module path: load_full_state_dict(reference parameters of fixture g2) -> the
    reference's 3 Adam steps on its batches split by rank -> full_state_dict() equals the reference's `after.*` arrays
    (trajectory tolerances of test_gpu_models.py::test_adam_trajectory_dense_exact), and the item table trained that way,
    served through model.index_corpus (both the local-rows and the routed form) -> the row-sharded mips_module ->
    model.forward(), returns the single-device model's top-K."""
    import torch.multiprocessing as mp
    _paths()
    if world != 1:
        world = resolve_world(world, backend)
    outdir = tempfile.mkdtemp()
    mp.spawn(_ckpt_worker, args=(world, _free_port(), outdir, backend), nprocs=world, join=True)
    g = np.load(os.path.join(HERE, "golden", "g2_base_aligned.npz"))
    for r in range(world):
        res = torch.load(os.path.join(outdir, f"ckpt{r}.pt"))
        assert np.allclose(res["losses"], g["adam_losses"], atol=1e-4), (res["losses"], g["adam_losses"])
        for k, v in res["sd"].items():
            after = torch.from_numpy(g["after." + k])
            assert v.shape == after.shape, k
            noise_only = float(np.abs(g["g." + k]).max()) < 1e-6
            err = (v - after).abs() - 1e-5 * after.abs()
            assert float(err.max()) <= 2 * 3 * 1e-3 * 1.05, (k, float(err.max()))
            if not noise_only:
                n_out = int((err > 5e-6).sum())
                assert n_out <= max(1, int(2e-3 * err.numel())) and float(err.max()) <= 2e-4, (k, n_out, float(err.max()))
        serve = torch.load(os.path.join(outdir, f"serve{r}.pt"))
        assert torch.equal(serve["idx"], serve["want"]), r
        # (the sharded model's own queries differ from the single-device ones by summation order: 1e-6; top-10 of 4096
        # random rows is not margin-gated here, so allow a swap of near-ties but not a different set)
        assert (serve["top_sharded"] == serve["want"]).float().mean() > 0.98, r


# ------------------------------------------------------------------ routing kernels
@pytest.mark.parametrize("n,n_rows,world", [(8192, 10_000_000, 8), (240, 307, 2), (50_000, 1_000_003, 7), (64, 64, 64),
                                            (204_800, 1_000_000, 8), (5000, 100_000, 1000)])
def test_route_kernels_match_cpu_restatement(n, n_rows, world):
    """tt_route_count / tt_route_build / tt_route_localize (csrc/route.hip) against the test double's torch
    restatement: bucket sizes and maximum, slot assignment (stable within an owner), padding, the inverse map,
    and the owner-side localisation with its sentinel."""
    _paths()
    from sharded_cpu_backend import OracleRouteKernels
    from two_tower_models_amd import parallel
    dev = torch.device("cuda:0")
    be, cpu = parallel._HipRouteKernels(dev), OracleRouteKernels()
    rpr = (n_rows + world - 1) // world
    g = torch.Generator().manual_seed(n)
    ids = torch.randint(0, n_rows, (n,), generator=g)
    ids[: n // 8] = ids[n // 8: 2 * (n // 8)]  # duplicates
    if world == 8:
        ids[-2000:] = torch.randint(3 * rpr, 4 * rpr, (2000,), generator=g)  # a lopsided bucket
    mx_d, mx_c = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32)
    pd = be.route_plan(ids.to(dev), n_rows, rpr, world, mx_d)
    pc = cpu.route_plan(ids, n_rows, rpr, world, mx_c)
    assert int(mx_d.item()) == int(mx_c.item())
    assert torch.equal(pd[3].cpu().long(), pc[2])  # bucket sizes
    cap = (int(mx_c.item()) + 63) // 64 * 64
    got = [t.cpu() for t in be.route_build(pd, rpr, world, cap)]
    want = cpu.route_build(pc, rpr, world, cap)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    send_ids, slot_of, src_of = got
    assert torch.equal(send_ids[slot_of], ids) and torch.equal(src_of[slot_of], torch.arange(n))
    assert int((send_ids >= 0).sum()) == n
    lo = 3 % world * rpr
    n_local = max(min(lo + rpr, n_rows) - lo, 0)
    loc = be.localize(send_ids.to(dev), lo, n_local).cpu()
    assert torch.equal(loc, cpu.localize(send_ids, lo, n_local))
    # the owner-side gather: rows for owned slots, zero rows for the sentinel and for padding (-1)
    table = torch.randn(max(n_local, 1), 32, generator=g)
    rows = be.gather_owned(table.to(dev), loc.to(dev), n_local).cpu()
    assert torch.equal(rows, cpu.gather_owned(table, loc, n_local))
    back = be.gather_owned(table.to(dev), src_of.to(dev).clamp(max=table.shape[0]), table.shape[0]).cpu()
    assert torch.equal(back, cpu.gather_owned(table, src_of.clamp(max=table.shape[0]), table.shape[0]))
    # the same stages for SEVERAL lookups per launch (tt_route_*_jobs: what a step uses): this list, a short second one and
    # a bf16-stored block, against the per-lookup results above
    ids2 = torch.randint(0, n_rows, (max(n // 7, 1),), generator=g)
    mx2 = torch.full((2,), -5, dtype=torch.int32, device=dev)  # written, not atomicMax'ed: garbage in is fine
    pl = be.route_plan_many([(ids.to(dev), n_rows, rpr, world), (ids2.to(dev), n_rows, rpr, world)], mx2)
    mc2 = torch.zeros(1, dtype=torch.int32)
    pc2 = cpu.route_plan(ids2, n_rows, rpr, world, mc2)
    assert mx2.cpu().tolist() == [int(mx_c.item()), int(mc2.item())]
    assert torch.equal(pl[0][3].cpu().long(), pc[2]) and torch.equal(pl[1][3].cpu().long(), pc2[2])
    cap2 = (int(mc2.item()) + 63) // 64 * 64
    built = be.route_build_many(pl, [rpr, rpr], world, [cap, cap2])
    for a, b in zip([t.cpu() for t in built[0]], want):
        assert torch.equal(a, b)
    for a, b in zip([t.cpu() for t in built[1]], cpu.route_build(pc2, rpr, world, cap2)):
        assert torch.equal(a, b)
    tb16 = table.to(torch.bfloat16)
    served = be.serve_many([(table.to(dev), send_ids.to(dev), lo, n_local), (tb16.to(dev), built[1][0], lo, n_local)])
    assert torch.equal(served[0][0].cpu(), loc) and torch.equal(served[0][1].cpu(), rows)
    loc2 = cpu.localize(built[1][0].cpu(), lo, n_local)
    assert torch.equal(served[1][0].cpu(), loc2) and torch.equal(served[1][1].cpu(), cpu.gather_owned(tb16, loc2, n_local))


# ------------------------------------------------------------------ results do not depend on host timing
@pytest.mark.parametrize("case,world", [("base_d128", 2), ("hist", 2)])
def test_sharded_step_is_independent_of_host_timing(case, world):
    """20 steps of the W = 2 module path (gloo ranks on cuda:0) with random 0-300 us sleeps after every enqueue, against the
    same run without them: losses and every parameter bit-identical on every rank."""
    import torch.multiprocessing as mp
    _paths()
    runs = []
    for seed in (None, 1234):
        outdir = tempfile.mkdtemp()
        mp.spawn(_worker, args=(world, _free_port(), outdir, case, "gloo", "torch", False, seed, 20), nprocs=world, join=True)
        runs.append([torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)])
    for r in range(world):
        assert runs[0][r]["losses"] == runs[1][r]["losses"], (runs[0][r]["losses"][-3:], runs[1][r]["losses"][-3:])
        for k, v in runs[0][r]["sd"].items():
            assert torch.equal(v, runs[1][r]["sd"][k]), (r, k)


def test_single_gpu_step_is_independent_of_host_timing():
    """The same at BASELINE config 2 on the single-GPU path (three streams: sweep, weight gradients / item tower, main):
    20 steps with the sleeps against 20 without, from the same seed -- bit-identical losses and tables."""
    _paths()
    import bench
    import two_tower_models_amd as A
    dev = torch.device("cuda:0")
    cfg = dict(bench.WORKLOADS["C2"])
    batches = bench.make_batches(cfg, 4, dev)

    def run(seed):
        model = bench.build_model(cfg, dev, seed=3)
        opt = A.DenseExactAdam(model.parameters(), lr=1e-3)
        restore = perturb_host_timing(seed) if seed is not None else (lambda: None)
        try:
            losses = []
            for i in range(20):
                loss = model.train_forward(*batches[i % 4])
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(loss.detach())
            torch.cuda.synchronize()
        finally:
            restore()
        return [float(l) for l in losses], {k: v.clone() for k, v in model.state_dict().items()}

    l0, sd0 = run(None)
    l1, sd1 = run(77)
    assert l0 == l1, (l0[-3:], l1[-3:])
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


# ------------------------------------------------------------------ the overlap STRUCTURE of the sharded step, without a second GPU
@pytest.mark.parametrize("transport", ["torch", "native"])
def test_exchanges_are_issued_before_and_waited_after_the_kernels_meant_to_cover_them(transport, monkeypatch):
    """World size 1 over RCCL with TT_COMM_FORCE_ASYNC (every exchange takes the asynchronous multi-GPU code path): the host
    ORDER of kernel launches, exchange issues and exchange waits of one train step -- _native.trace -- shows the cover each
    exchange was designed to have, so that the first 8-GPU run only has to confirm numbers:
      * every lookup's rows are in flight before the first tower kernel;
      * the item-embedding all-gather is issued after the item tower, BEFORE the user tower's forward kernel, and waited after it;
      * the dI reduce-scatter is issued before the user tower's backward kernel and waited only at the item tower's;
      * row gradients travel until the optimiser's step(), the dense all-reduce under the table finish."""
    import torch.distributed as dist
    _paths()
    import two_tower_models_amd as A
    from two_tower_models_amd import _native as N
    from two_tower_models_amd import collectives, parallel
    monkeypatch.setenv("TT_COMM_FORCE_ASYNC", "1")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        if transport == "native":
            from two_tower_models_amd.comm import NativeComm
            collectives.use_native_transport(NativeComm.from_torch_distributed(dev))
        model = build_case("base_d128").to(dev)
        parallel.shard_model_(model)
        opt = A.DenseExactAdam(model.parameters(), lr=1e-3)
        batches = [tuple(t.to(dev) for t in b) for b in make_batches("base_d128", 0, 3)]
        for i, b in enumerate(batches):
            if i == 2:
                N.trace = []
            loss = model.train_forward(*b)
            if i + 1 < len(batches):
                parallel.plan_ahead(model._lookup_plan(batches[i + 1][0], batches[i + 1][2], batches[i + 1][3]))
            opt.zero_grad()
            loss.backward()
            opt.step()
        tr, N.trace = N.trace, None
        torch.cuda.synchronize()
        where = lambda name: [i for i, n in enumerate(tr) if n == name]
        fwd, bwd = where("tt_tower_fwd_x"), where("tt_tower_bwd_data_x")
        assert len(fwd) == 2 and len(bwd) == 2, tr  # forward: item tower, user tower; backward: user tower, item tower
        rows_issued = where("issue:lookup_rows_alltoall")
        assert len(rows_issued) == 2 and max(rows_issued) < fwd[0], tr
        (ag_issue,), (ag_wait,) = where("issue:item_emb_allgather"), where("wait:item_emb_allgather")
        assert fwd[0] < ag_issue < fwd[1] < ag_wait, (fwd, ag_issue, ag_wait)
        (rs_issue,), (rs_wait,) = where("issue:dI_reduce_scatter"), where("wait:dI_reduce_scatter")
        assert rs_issue < bwd[0] < rs_wait < bwd[1], (bwd, rs_issue, rs_wait)
        (pack,), (finish,) = where("tt_pack_grads"), where("tt_adam_tables_finish")
        grads_issued, grads_waited = where("issue:rowgrad_alltoall"), where("wait:rowgrad_alltoall")
        assert len(grads_issued) == 2 and len(grads_waited) == 2
        assert max(grads_issued) < pack < min(grads_waited) and max(grads_waited) < finish, tr  # waited for in step() only
        (ar_issue,), (ar_wait,) = where("issue:dense_grad_allreduce"), where("wait:dense_grad_allreduce")
        assert pack < ar_issue < finish < ar_wait, (pack, ar_issue, finish, ar_wait)
        assert not parallel._DEFERRED
    finally:
        N.trace = None
        collectives.use_native_transport(None)
        dist.destroy_process_group()


# ------------------------------------------------------------------ the RCCL code paths on the 1-GPU box
@pytest.mark.parametrize("transport", ["torch", "native"])
def test_rccl_async_paths_at_world1_with_the_real_message_sizes(transport, monkeypatch):
    """The code a multi-GPU node runs -- `*_start` / `.wait()` through RCCL's async collectives on the process group's
    stream (transport torch) and through tt_comm_* on the communication stream (transport native) -- executed on the
    1-GPU box: TT_COMM_FORCE_ASYNC takes those paths at world size 1, where every collective is the identity, at the
    message sizes of the P step at W = 8 (ids [8 x cap] int64, rows [8 x cap, 128] fp32 = 8 MB, item embeddings 4 MB,
    the 0.55 MB dense-gradient buffer, scalars).  Then three whole steps of the sharded MODULE path over the same paths
    (incl. the deferred reduce-scatter wait) against the oracle, with the per-exchange timing on, and bit-identical to
    the synchronous run."""
    import torch.distributed as dist
    _paths()
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    from two_tower_models_amd import collectives, parallel
    monkeypatch.setenv("TT_COMM_FORCE_ASYNC", "1")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        if transport == "native":
            from two_tower_models_amd.comm import NativeComm
            collectives.use_native_transport(NativeComm.from_torch_distributed(dev))
        g = torch.Generator(device=dev).manual_seed(5)
        cap = 1088
        ids = torch.randint(0, 10_000_000, (8 * cap,), device=dev, generator=g)
        rows = torch.randn(8 * cap, 128, device=dev, generator=g)
        emb = torch.randn(8192, 128, device=dev, generator=g)
        flat = torch.randn(136_192, device=dev, generator=g)
        collectives.comm_timing(True)
        side = torch.randn(4096, 4096, device=dev, generator=g)
        p_ids = collectives.all_to_all_rows_start(ids, tag="ids")
        p_rows = collectives.all_to_all_rows_start(rows, tag="rows")
        p_ag = collectives.all_gather_rows_start(emb, tag="ag")
        p_rs = collectives.reduce_scatter_rows_start(emb, tag="rs")
        f2 = flat.clone()
        p_ar = collectives.all_reduce_start_(f2, tag="ar")
        busy = side @ side  # compute queued between start and wait: the exchanges run underneath it
        assert collectives._rccl_async(rows) or collectives._native(rows)
        assert torch.equal(p_ids.wait(), ids) and torch.equal(p_rows.wait(), rows)
        assert torch.equal(p_ag.wait(), emb) and torch.equal(p_rs.wait(), emb) and torch.equal(p_ar.wait(), flat)
        k = torch.tensor([7, 3, 9], dtype=torch.int32, device=dev)
        assert collectives.all_reduce_start_(k, op=dist.ReduceOp.MAX, tag="caps").wait().tolist() == [7, 3, 9]
        summ = collectives.comm_timing_summary(1)
        assert set(summ) == {"ids", "rows", "ag", "rs", "ar", "caps"} and all(v["span_ms"] >= v["exposed_ms"] >= 0 for v in summ.values())
        if transport == "native":
            assert all("wire_ms" in v for v in summ.values())
        assert float(busy.abs().sum()) > 0

        # whole steps over the same paths
        def run():
            model = build_case("base_d128").to(dev)
            parallel.shard_model_(model)
            opt = A.DenseExactAdam(model.parameters(), lr=1e-3)
            batches = [tuple(t.to(dev) for t in b) for b in make_batches("base_d128", 0, 3)]
            losses = []
            for i, b in enumerate(batches):
                loss = model.train_forward(*b)
                if i + 1 < len(batches):
                    parallel.plan_ahead(model._lookup_plan(batches[i + 1][0], batches[i + 1][2], batches[i + 1][3]))
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(float(loss))
            return losses, {k: v.clone() for k, v in model.state_dict().items()}, batches

        collectives.comm_timing(True)
        got, sd, batches = run()
        per_step = collectives.comm_timing_summary(3)
        collectives.comm_timing(False)
        assert {"lookup_ids_alltoall", "lookup_rows_alltoall", "rowgrad_alltoall", "dense_grad_allreduce",
                "item_emb_allgather", "dI_reduce_scatter"} <= set(per_step)
        assert not parallel._DEFERRED  # every deferred reduce-scatter was waited for by its consumer
        params = {k: v.detach().clone() for k, v in build_case("base_d128").state_dict().items()}
        state = R.AdamState(params)
        want = [R.train_step(params, state, [t.cpu() for t in b], torch.tensor([UVW])) for b in batches]
        assert np.allclose(got, want, atol=1e-4), (got, want)
        # ... and the forced-async run is the synchronous run, bit for bit (same kernels, same order; only where the
        # collectives execute differs)
        monkeypatch.delenv("TT_COMM_FORCE_ASYNC")
        collectives.use_native_transport(None)
        got2, sd2, _ = run()
        assert got2 == got and all(torch.equal(sd[k], sd2[k]) for k in sd)
    finally:
        collectives.comm_timing(False)
        collectives.use_native_transport(None)
        dist.destroy_process_group()


def test_native_comm_world1_every_collective():
    """tt_comm_* of the C ABI (csrc/comm.cpp, RCCL bound at run time) with a one-rank communicator: id, init,
    size, and every collective parallel.py uses -- at world size 1 each is the identity, which checks the binding,
    the dtype / op mapping and the stream plumbing.  World sizes > 1 run in the nccl cases above on multi-GPU nodes."""
    from two_tower_models_amd import _native as N
    from two_tower_models_amd.comm import NativeComm
    dev = torch.device("cuda:0")
    c = NativeComm(NativeComm.unique_id(), 0, 1, dev)
    try:
        assert c.size() == (0, 1)
        x = torch.randn(64, 128, device=dev)
        ids = torch.arange(640, device=dev)
        side = torch.cuda.Stream()
        assert torch.equal(c.all_to_all(x), x) and torch.equal(c.all_to_all(ids), ids)
        assert torch.equal(c.all_gather(x), x) and torch.equal(c.reduce_scatter(x), x)
        y = x.clone()
        assert torch.equal(c.all_reduce_(y), x) and torch.equal(c.all_reduce_(y, N.TT_COMM_MAX), x)
        k = torch.tensor([7, 3, 9], dtype=torch.int32, device=dev)
        assert c.all_reduce_(k, N.TT_COMM_MAX).tolist() == [7, 3, 9]
        assert torch.equal(c.broadcast_(y, 0), x)
        side.wait_stream(torch.cuda.current_stream())
        z = c.all_to_all(x, stream=side)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(z, x)
        with pytest.raises(TypeError):
            c.all_gather(x.double())
        with pytest.raises(RuntimeError, match="in-place"):
            c.all_to_all(x, recv=x)
    finally:
        c.close()


def test_train_py_world_size_2_runs_the_reference_loop():
    """`torchrun --nproc_per_node 2 -m two_tower_models_amd.train --world_size 2` (gloo hook: both ranks on cuda:0): the
    reference's script surface (ref:train/train.py:138-183) on row-sharded tables -- prints the reference's epoch lines
    from rank 0 only, the loss goes down."""
    import re
    import subprocess
    env = dict(os.environ, TT_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), "-m", "two_tower_models_amd.train", "--world_size", "2",
                        "--num_epochs", "3", "--num_samples", "2048", "--batch_size", "128", "--embedding_dim", "32",
                        "--model", "debias", "--learning_rate", "0.01", "--dtype", "bf16", "--retrieve"],
                       cwd=os.path.dirname(HERE), env=env, capture_output=True, text=True, timeout=900, stdin=subprocess.DEVNULL)
    assert r.returncode == 0, r.stderr[-3000:]
    losses = [float(m) for m in re.findall(r"Epoch \[\d/3\] - Loss: ([0-9.]+)", r.stdout)]
    assert len(losses) == 3 and losses[-1] < losses[0], r.stdout[-1000:]
    assert r.stdout.count("Running on device") == 1
    # --retrieve: model.index_corpus + model.forward() on the row-sharded model (SURVEY section 5's --dtype: bf16 corpus blocks)
    assert "Retrieved top-10 of 1024 items for 128 users (bfloat16 corpus, 2 row blocks)" in r.stdout, r.stdout[-600:]
