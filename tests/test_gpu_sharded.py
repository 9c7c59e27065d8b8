"""ShardedTrainer with the PRODUCT backend (HipBackend, libtt_hotpath.so) on one MI355X:
world_size 1 over RCCL must reproduce the oracle's train steps (loss trajectory, tables,
dense parameters).  The W > 1 routing logic is covered on CPU by tests/test_sharded_cpu.py;
this test covers the HIP arithmetic behind the same interface."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("cfg", [dict(n_users=300, n_items=500, D=128, F=8, B=256, H=2),
                                 dict(n_users=53, n_items=71, D=40, F=20, B=32, H=2)])
def test_world1_hip_backend_matches_oracle(cfg):
    import torch.distributed as dist
    from oracle import cpu_ref as R
    from test_sharded_cpu import _dense_init
    from two_tower_models_amd import sharded
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        dense = _dense_init(cfg)
        tr = sharded.ShardedTrainer(cfg, dev, negatives="global", user_value_weights=(0.7,), dense_init=dense)
        assert isinstance(tr.be, sharded.HipBackend)
        params = dict(dense)
        params["user_id_embedding_arch.weight"] = tr.users.weight.cpu().clone()
        params["item_id_embedding_arch.weight"] = tr.items.weight.cpu().clone()
        state = R.AdamState(params)
        batches = tr.make_batches(3, seed=7)
        got, want = [], []
        for b in batches:
            got.append(float(tr.step(b)))
            want.append(R.train_step(params, state, [t.cpu() for t in b], torch.tensor([0.7])))
        assert np.allclose(got, want, atol=1e-4), (got, want)
        assert torch.allclose(tr.users.weight.cpu(), params["user_id_embedding_arch.weight"], atol=5e-6)
        assert torch.allclose(tr.items.weight.cpu(), params["item_id_embedding_arch.weight"], atol=5e-6)
        for k, v in tr.params.items():
            noise_only = k in ("item_tower_arch.bias", "item_features_arch.2.bias")
            assert torch.allclose(v.cpu(), params[k], atol=6.6e-3 if noise_only else 5e-6), k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("split", [False, True, "kept"])
def test_world1_split_fp16_logits_match_oracle_at_unchanged_tolerances(monkeypatch, split):
    """TT_CE_F16X2 (exploratory): the trainer with the split-fp16 logits pair (csrc/ce_f16x2.hip) against the oracle's train
    steps at the SAME criterion as the fp32-MFMA pair (split = False runs that one through the identical assertions) --
    loss 1e-4, tables and dense parameters 5e-6 after three Adam steps -- at a shape the pair takes (B = 1024, D = 128),
    and the pair is what ran (split = True: its default form, no logits buffer; "kept": its first form, TT_CE16_KEEP)."""
    import torch.distributed as dist
    from oracle import cpu_ref as R
    from test_sharded_cpu import _dense_init
    from two_tower_models_amd import sharded
    cfg = dict(n_users=3000, n_items=5000, D=128, F=8, B=1024, H=2)
    dev = torch.device("cuda:0")
    monkeypatch.setattr(sharded, "_CE_F16X2", bool(split))
    monkeypatch.setattr(sharded, "_CE16_KEEP", split == "kept")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        dense = _dense_init(cfg)
        tr = sharded.ShardedTrainer(cfg, dev, negatives="global", user_value_weights=(0.7,), dense_init=dense)
        calls = []
        which = "tt_ce16_bwd_kept" if split == "kept" else "tt_ce16_bwd_recompute"
        real = getattr(tr.be.lib, which)
        monkeypatch.setattr(tr.be.lib, which, lambda *a: (calls.append(1), real(*a))[1])
        params = dict(_dense_init(cfg))
        params["user_id_embedding_arch.weight"] = tr.users.weight.cpu().clone()
        params["item_id_embedding_arch.weight"] = tr.items.weight.cpu().clone()
        state = R.AdamState(params)
        batches = tr.make_batches(3, seed=7)
        got, want = [], []
        for b in batches:
            got.append(float(tr.step(b)))
            want.append(R.train_step(params, state, [t.cpu() for t in b], torch.tensor([0.7])))
        assert len(calls) == (3 if split else 0)
        assert np.allclose(got, want, atol=1e-4), (got, want)
        # Adam's first steps turn rounding noise on near-zero gradients into +-lr (lr * g / (|g| + eps)): at this batch size
        # a handful of elements per tensor land beyond 5e-6 whichever kernels ran (split = False: 1 failed on the strict
        # form) -- so the criterion, identical for both, is 5e-6 for all but 0.2 % of a tensor's elements and 2.1 lr for those
        tensors = {"users": (tr.users.weight.cpu(), params["user_id_embedding_arch.weight"]),
                   "items": (tr.items.weight.cpu(), params["item_id_embedding_arch.weight"])}
        tensors.update({k: (v.cpu(), params[k]) for k, v in tr.params.items()})
        for k, (v, want_v) in tensors.items():
            if k in ("item_tower_arch.bias", "item_features_arch.2.bias"):  # analytically zero gradient: rounding noise only
                assert torch.allclose(v, want_v, atol=6.6e-3), k
                continue
            d = (v - want_v).abs()
            bad = int((d > 5e-6).sum())
            assert bad <= max(2, v.numel() // 500) and float(d.max()) <= 3 * 2.1e-3, (k, bad, v.numel(), float(d.max()))
    finally:
        dist.destroy_process_group()


def test_mips_merge_kernel_and_world1_sharded_mips():
    """tt_mips_merge on hand-made shard lists (ties across shards, padding) and ShardedMIPS with
    the product backend at world size 1."""
    import torch.distributed as dist
    import fixture_gen as fg
    from oracle import cpu_ref as R
    from two_tower_models_amd import ops, sharded
    dev = torch.device("cuda:0")
    # 3 "shards" x K=4 candidates for 2 queries; equal scores across shards must order by index
    sc = torch.tensor([[5., 3., 3., 1., 5., 4., 3., 0., 9., 3., 2., 0.],
                       [1., 1., 1., 1., 1., 1., 1., 0., 1., 1., 0., 0.]])
    ix = torch.tensor([[10, 11, 12, 13, 20, 21, 22, -1, 3, 30, 31, -1],
                       [7, 8, 9, 10, 1, 2, 3, -1, 4, 5, -1, -1]])
    oi, os_ = ops.mips_merge(sc.to(dev), ix.to(dev), 5)
    assert oi.cpu().tolist() == [[3, 10, 20, 21, 11], [1, 2, 3, 4, 5]]
    assert os_.cpu().tolist() == [[9., 5., 5., 4., 3.], [1., 1., 1., 1., 1.]]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=dev)
    try:
        corpus = torch.from_numpy(fg.exact_mips_corpus(5000, 64))
        q = torch.from_numpy(fg.exact_mips_queries(9, 64))
        m = sharded.ShardedMIPS(corpus.to(dev), 0)
        idx, s = m.search(q.to(dev), 100)
        want_idx, want_sc, _ = R.mips_topk(q, corpus, 100)
        assert torch.equal(idx.cpu(), want_idx) and torch.equal(s.cpu(), want_sc)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ world size > 1 on ONE GPU
# RCCL refuses two ranks on the same device, so these run the ranks as separate processes that
# share cuda:0 and exchange through gloo (sharded.py stages device tensors through the host for
# gloo).  Everything except the transport is the product path: HipBackend kernels with non-zero
# diag_offset, zero rows for ids another rank owns, sentinel rows in the Adam plan, the side-stream
# sweep, mips_merge over W candidate lists.
MULTI_CFGS = {"d128": dict(n_users=300, n_items=500, D=128, F=8, B=128, H=2),
              "ragged": dict(n_users=53, n_items=71, D=40, F=20, B=24, H=2),
              # TwoTowerWithUserHistoryEncoder: 4 heads x dh 32 (MFMA attention), B*H = 240 history rows per rank
              "hist": dict(n_users=211, n_items=307, D=128, F=8, B=40, H=6, model="hist"),
              # a batch larger than either table (every row looked up several times, by several ranks), tables that do
              # not divide by the world size, fused-tower width 64
              "crowded": dict(n_users=37, n_items=29, D=64, F=5, B=96, H=3),
              # history model with H = 50 (the BASELINE length: attention tiles padded to 64) and a table smaller than
              # one rank's history list
              "hist50": dict(n_users=90, n_items=131, D=128, F=8, B=24, H=50, model="hist"),
              # an item table with fewer rows than ranks x rows-per-rank: the last rank owns NO item row (found by
              # tools/fuzz_sharded.py: the lookups and the table Adam of an empty row block used to be rejected)
              "empty_block": dict(n_users=338, n_items=9, D=64, F=20, B=33, H=1, model="hist")}
MULTI_STEPS = 3


def _cfg(name):
    """A named case, or any configuration as "json:{...}" (tools/fuzz_sharded.py; the spawned ranks re-import this
    module, so the configuration has to travel in the name)."""
    import json
    return json.loads(name[5:]) if name.startswith("json:") else MULTI_CFGS[name]


def _multi_init(cfg):
    """Dense init with small tower weights + 0.5-scaled tables: logits O(1), loss ~ 3.  (With O(100)
    logits most rows are saturated, p - 1 cancels catastrophically and whole rows carry a 1e-2
    relative gradient error in ANY fp32 implementation.)"""
    from test_sharded_cpu import _dense_init, _hist_dense_init
    init = _hist_dense_init if cfg.get("model") == "hist" else _dense_init
    hist = cfg.get("model") == "hist"
    dense = {k: ((0.05 if hist else 0.15) * v if k.endswith("tower_arch.weight") else v) for k, v in init(cfg).items()}
    if hist:  # D = 128: keep the attention logits and the summary O(1) as well
        dense = {k: (0.3 * v if ("in_proj_weight" in k or "out_proj.weight" in k) else v) for k, v in dense.items()}
    g = torch.Generator().manual_seed(6)
    ut = 0.5 * torch.randn(cfg["n_users"], cfg["D"], generator=g)
    it = 0.5 * torch.randn(cfg["n_items"], cfg["D"], generator=g)
    return dense, ut, it


def _init_pg(backend, rank, world, port):
    """gloo: every rank on cuda:0 (1-GPU test box).  nccl (= RCCL): one rank per device."""
    import torch.distributed as dist
    dev = torch.device(f"cuda:{rank}" if backend == "nccl" else "cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        import os
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    return dev


def _multi_worker(rank, world, port, outdir, cfg_name, negatives, backend="gloo", routing="alltoall", transport="torch"):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from two_tower_models_amd import sharded
    cfg = _cfg(cfg_name)
    dense, ut, it = _multi_init(cfg)
    dev = _init_pg(backend, rank, world, port)
    try:
        tr = sharded.ShardedTrainer(cfg, dev, negatives=negatives, user_value_weights=(0.7,),
                                    dense_init=dense, routing=routing, transport=transport)
        assert isinstance(tr.be, sharded.HipBackend)
        assert (sharded._NATIVE is not None) == (transport == "native")
        tr.users.weight[: tr.users.hi - tr.users.lo].copy_(ut[tr.users.lo:tr.users.hi])
        tr.items.weight[: tr.items.hi - tr.items.lo].copy_(it[tr.items.lo:tr.items.hi])
        batches = tr.make_batches(MULTI_STEPS, seed=99)
        # step 0 announces batch 1 (routes planned one step ahead); batch 2 arrives unannounced
        losses = [float(tr.step(b, batches[i + 1] if i == 0 else None)) for i, b in enumerate(batches)]
        torch.cuda.synchronize()
        torch.save({"losses": losses, "users": tr.users.weight.cpu(), "items": tr.items.weight.cpu(),
                    "lo_hi": (tr.users.lo, tr.users.hi, tr.items.lo, tr.items.hi),
                    "dense": {k: v.cpu() for k, v in tr.params.items()},
                    "batches": [tuple(t.cpu() for t in b) for b in batches]},
                   os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _resolve_world(world, backend):
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


@pytest.mark.parametrize("world,cfg_name,backend,routing,transport",
                         [(2, "d128", "gloo", "alltoall", "torch"), (3, "ragged", "gloo", "alltoall", "torch"),
                          (2, "hist", "gloo", "alltoall", "torch"), (2, "d128", "gloo", "allgather", "torch"),
                          (2, "hist", "gloo", "allgather", "torch"),
                          (3, "crowded", "gloo", "alltoall", "torch"), (3, "crowded", "gloo", "allgather", "torch"),
                          (2, "hist50", "gloo", "alltoall", "torch"), (3, "hist50", "gloo", "alltoall", "torch"),
                          (4, "empty_block", "gloo", "alltoall", "torch"), (4, "empty_block", "gloo", "allgather", "torch"),
                          # RCCL, one device per rank: skipped on the 1-GPU test boxes, run on any multi-GPU node
                          (2, "d128", "nccl", "alltoall", "torch"), ("all", "d128", "nccl", "alltoall", "torch"),
                          ("all", "ragged", "nccl", "alltoall", "torch"), (2, "hist", "nccl", "alltoall", "torch"),
                          ("all", "d128", "nccl", "allgather", "torch"),
                          # the C ABI's own collectives (tt_comm_*) instead of torch's process group
                          (2, "d128", "nccl", "alltoall", "native"), ("all", "hist", "nccl", "alltoall", "native"),
                          ("all", "d128", "nccl", "allgather", "native")])
def test_multi_rank_hip_backend_equals_reference_on_concatenated_batch(world, cfg_name, backend, routing, transport,
                                                                       outlier_frac=2e-3):
    import os
    import tempfile
    import torch.multiprocessing as mp
    from oracle import cpu_ref as R
    world = _resolve_world(world, backend)
    cfg = _cfg(cfg_name)
    outdir = tempfile.mkdtemp()
    mp.spawn(_multi_worker, args=(world, _free_port(), outdir, cfg_name, "global", backend, routing, transport),
             nprocs=world, join=True)
    res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    dense, ut, it = _multi_init(cfg)
    params = dict(dense)
    params["user_id_embedding_arch.weight"] = ut.clone()
    params["item_id_embedding_arch.weight"] = it.clone()
    state = R.AdamState(params)
    want = []
    kw = {}
    if cfg.get("model") == "hist":
        kw = dict(with_history=True, heads=4, pos_table=R.positional_table(cfg["H"], cfg["D"]))
    for s in range(MULTI_STEPS):
        cat = [torch.cat([res[r]["batches"][s][k] for r in range(world)]) for k in range(7)]
        want.append(R.train_step(params, state, cat, torch.tensor([0.7]), **kw))
    for r in range(world):
        assert np.allclose(res[r]["losses"], want, atol=1e-4), (res[r]["losses"], want)
        ulo, uhi, ilo, ihi = res[r]["lo_hi"]
        # Adam's early updates are lr * g / (|g| + eps)-like: the few elements whose gradient happens
        # to be ~1e-4 of the typical size turn a 1e-7 relative summation-order difference into a
        # ~1e-5 step difference (the CPU restatement run as 2 ranks shows the same outliers against
        # the 1-rank reference).  So: all but <= 0.2% of the elements tight, every element within
        # the steps * lr bound.
        named = [("users", res[r]["users"][: uhi - ulo], params["user_id_embedding_arch.weight"][ulo:uhi]),
                 ("items", res[r]["items"][: ihi - ilo], params["item_id_embedding_arch.weight"][ilo:ihi])]
        named += [(k, v, params[k]) for k, v in res[r]["dense"].items()]
        for name, got, ref in named:
            err = (got - ref).abs()
            if err.numel() == 0:  # a rank that owns no row of a table (fewer rows than ranks)
                continue
            assert float(err.max()) <= 2.2e-3 * MULTI_STEPS, (name, r, float(err.max()))
            # zero true gradient, noise only: the item-side biases and the key third of every in_proj_bias
            noise_only = name in ("item_tower_arch.bias", "item_features_arch.2.bias") or name.endswith("in_proj_bias")
            if not noise_only:  # all but 0.2 % of the elements (at least two: a bias has 128-256 of them) within 5e-6
                # (two, not one: tools/fuzz_sharded.py seed 32 case 24 -- W = 3, B = 8 per rank, D = 24 -- has two such elements
                # in a 256-element bias; the round-2 tree shows the identical finding, so it is the lottery described above)
                assert int((err > 5e-6).sum()) <= max(2, int(outlier_frac * err.numel())), (name, r, int((err > 5e-6).sum()), err.numel())
        # replicas stay bit-identical
        assert all(torch.equal(v, res[0]["dense"][k]) for k, v in res[r]["dense"].items())


def _multi_mips_worker(rank, world, port, outdir, C, K, backend="gloo"):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import fixture_gen as fg
    from two_tower_models_amd import sharded
    dev = _init_pg(backend, rank, world, port)
    try:
        corpus = torch.from_numpy(fg.exact_mips_corpus(C, 64))
        lo, hi = sharded.ShardedMIPS.block_range(C, rank, world)
        m = sharded.ShardedMIPS(corpus[lo:hi].to(dev), lo)
        q = torch.from_numpy(fg.exact_mips_queries(6 * world, 64))[rank * 6:(rank + 1) * 6]
        idx, sc = m.search(q.to(dev), K)
        torch.save({"idx": idx.cpu(), "sc": sc.cpu()}, os.path.join(outdir, f"mips{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,C,K,backend", [(2, 9000, 100, "gloo"), (3, 200, 80, "gloo"),
                                               (4, 5, 3, "gloo"),  # the last rank's corpus block is empty (tools/fuzz_sharded_mips.py)
                                               (2, 9000, 100, "nccl"), ("all", 9000, 100, "nccl")])
def test_multi_rank_sharded_mips_hip_backend(world, C, K, backend):
    import os
    import tempfile
    import torch.multiprocessing as mp
    import fixture_gen as fg
    from oracle import cpu_ref as R
    world = _resolve_world(world, backend)
    outdir = tempfile.mkdtemp()
    mp.spawn(_multi_mips_worker, args=(world, _free_port(), outdir, C, K, backend), nprocs=world, join=True)
    corpus = torch.from_numpy(fg.exact_mips_corpus(C, 64))
    q = torch.from_numpy(fg.exact_mips_queries(6 * world, 64))
    want_idx, want_sc, _ = R.mips_topk(q, corpus, K)
    for r in range(world):
        got = torch.load(os.path.join(outdir, f"mips{r}.pt"))
        assert torch.equal(got["idx"], want_idx[r * 6:(r + 1) * 6])
        assert torch.equal(got["sc"], want_sc[r * 6:(r + 1) * 6])


# ------------------------------------------------------------------ SURVEY 8f-4 under the product backend
def _ckpt_worker(rank, world, port, outdir, backend):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from two_tower_models_amd import sharded
    g = np.load(os.path.join(here, "golden", "g2_base_aligned.npz"))
    n_users, du, iu, n_items, di, ii, Tn, B, H = (int(v) for v in g["cfg"])
    dev = _init_pg(backend, rank, world, port)
    try:
        Bl = B // world
        cfg = dict(n_users=n_users, n_items=n_items, D=du, F=iu, B=Bl, H=H)
        tr = sharded.ShardedTrainer(cfg, dev, negatives="global", user_value_weights=tuple(float(v) for v in g["uvw"]))
        assert isinstance(tr.be, sharded.HipBackend)
        # a REFERENCE-format checkpoint (the parameters the reference model was created with) into the shards ...
        tr.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p.")})
        names = ("user_id", "user_features", "user_history", "item_id", "item_features", "position", "labels")
        losses = []
        for s in range(3):  # ... the reference's three batches, split by rank ...
            b = [torch.from_numpy(g[f"step{s}.in.{n}"])[rank * Bl:(rank + 1) * Bl].to(dev) for n in names]
            losses.append(float(tr.step(b)))
        sd = tr.state_dict()  # ... and back out under the reference's Parameter names
        # serve the trained item table: this rank's catalogue block through the item tower -> ShardedMIPS
        feats = torch.from_numpy(g["step0.in.item_features"])  # any [*, II] features: row r of the catalogue gets row r % B
        cat_feats = feats[torch.arange(n_items) % B]
        mips = tr.index_corpus(cat_feats[tr.items.lo:tr.items.hi])
        torch.save({"losses": losses, "sd": {k: v.cpu() for k, v in sd.items()}, "cat_feats": cat_feats},
                   os.path.join(outdir, f"ckpt{rank}.pt"))
        dist.barrier()
        # queries: the user embeddings of the trained model, from a single-device module fed the gathered state
        import two_tower_models_amd as A
        single = A.TwoTowerBaseRetrieval(10, n_users, du, iu, n_items, di, ii, [float(v) for v in g["uvw"]],
                                         A.BaselineMIPSModule(corpus_size=n_items, embedding_dim=di))
        single.load_state_dict(sd)
        single = single.to(dev)
        with torch.no_grad():
            single.index_corpus(torch.arange(n_items, device=dev), cat_feats.to(dev))
            users = [torch.from_numpy(g[f"step2.in.{n}"])[rank * Bl:(rank + 1) * Bl].to(dev) for n in names[:3]]
            want_top = single(*users)
            q = single.compute_user_embedding(*users)
        idx, _ = mips.search(q, 10)
        torch.save({"idx": idx.cpu(), "want": want_top.cpu()}, os.path.join(outdir, f"serve{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(1, "nccl"), (2, "gloo"), (2, "nccl"), ("all", "nccl")])
def test_sharded_checkpoint_adaptor_and_corpus_serving_hip_backend(world, backend):
    """SURVEY 8f item 4 on the product backend: ShardedTrainer.load_state_dict(reference parameters of fixture g2)
    -> the reference's 3 Adam steps on its batches split by rank -> state_dict() equals the reference's `after.*`
    arrays (trajectory tolerances of test_gpu_models.py::test_adam_trajectory_dense_exact), and the item table
    trained that way, served through index_corpus -> ShardedMIPS, returns the single-device model's top-K."""
    import os
    import tempfile
    import torch.multiprocessing as mp
    if world != 1:
        world = _resolve_world(world, backend)
    outdir = tempfile.mkdtemp()
    mp.spawn(_ckpt_worker, args=(world, _free_port(), outdir, backend), nprocs=world, join=True)
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "g2_base_aligned.npz"))
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


@pytest.mark.parametrize("n,n_rows,world", [(8192, 10_000_000, 8), (240, 307, 2), (50_000, 1_000_003, 7), (64, 64, 64),
                                            (204_800, 1_000_000, 8), (5000, 100_000, 1000)])
def test_route_kernels_match_cpu_restatement(n, n_rows, world):
    """tt_route_count / tt_route_build / tt_route_localize (csrc/route.hip) against the test double's torch
    restatement: bucket sizes and maximum, slot assignment (stable within an owner), padding, the inverse map,
    and the owner-side localisation with its sentinel."""
    from sharded_cpu_backend import OracleBackend
    from two_tower_models_amd import sharded
    dev = torch.device("cuda:0")
    be, cpu = sharded.HipBackend(dev), OracleBackend()
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


@pytest.mark.parametrize("transport", ["torch", "native"])
def test_rccl_async_paths_at_world1_with_the_real_message_sizes(transport, monkeypatch):
    """The code a multi-GPU node runs -- `*_start` / `.wait()` through RCCL's async collectives on the process group's
    stream (transport torch) and through tt_comm_* on the communication stream (transport native) -- executed on the
    1-GPU box: TT_COMM_FORCE_ASYNC takes those paths at world size 1, where every collective is the identity, at the
    message sizes of the P step at W = 8 (ids [8 x cap] int64, rows [8 x cap, 128] fp32 = 8 MB, item embeddings 4 MB,
    the 0.55 MB dense-gradient buffer, scalars).  Then three whole steps of the routed trainer over the same paths
    against the oracle, with the per-exchange timing on."""
    import torch.distributed as dist
    from oracle import cpu_ref as R
    from test_sharded_cpu import _dense_init
    from two_tower_models_amd import sharded
    monkeypatch.setenv("TT_COMM_FORCE_ASYNC", "1")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        cfg = dict(n_users=3000, n_items=5000, D=128, F=8, B=512, H=2)
        dense = _dense_init(cfg)
        tr = sharded.ShardedTrainer(cfg, dev, negatives="global", user_value_weights=(0.7,), dense_init=dense,
                                    transport=transport)
        assert tr.transport == transport and (sharded._NATIVE is not None) == (transport == "native")
        g = torch.Generator(device=dev).manual_seed(5)
        cap = 1088
        ids = torch.randint(0, 10_000_000, (8 * cap,), device=dev, generator=g)
        rows = torch.randn(8 * cap, 128, device=dev, generator=g)
        emb = torch.randn(8192, 128, device=dev, generator=g)
        flat = torch.randn(136_192, device=dev, generator=g)
        sharded.comm_timing(True)
        side = torch.randn(4096, 4096, device=dev, generator=g)
        p_ids = sharded.all_to_all_rows_start(ids, tag="ids")
        p_rows = sharded.all_to_all_rows_start(rows, tag="rows")
        p_ag = sharded.all_gather_rows_start(emb, tag="ag")
        p_rs = sharded.reduce_scatter_rows_start(emb, tag="rs")
        f2 = flat.clone()
        p_ar = sharded.all_reduce_start_(f2, tag="ar")
        busy = side @ side  # compute queued between start and wait: the exchanges run underneath it
        assert sharded._rccl_async(rows) or sharded._native(rows)
        assert torch.equal(p_ids.wait(), ids) and torch.equal(p_rows.wait(), rows)
        assert torch.equal(p_ag.wait(), emb) and torch.equal(p_rs.wait(), emb) and torch.equal(p_ar.wait(), flat)
        k = torch.tensor([7, 3, 9], dtype=torch.int32, device=dev)
        assert sharded.all_reduce_start_(k, op=dist.ReduceOp.MAX, tag="caps").wait().tolist() == [7, 3, 9]
        summ = sharded.comm_timing_summary(1)
        assert set(summ) == {"ids", "rows", "ag", "rs", "ar", "caps"} and all(v["span_ms"] >= v["exposed_ms"] >= 0 for v in summ.values())
        if transport == "native":
            assert all("wire_ms" in v for v in summ.values())
        assert float(busy.abs().sum()) > 0
        # whole steps over the same paths
        params = dict(dense)
        params["user_id_embedding_arch.weight"] = tr.users.weight.cpu().clone()
        params["item_id_embedding_arch.weight"] = tr.items.weight.cpu().clone()
        state = R.AdamState(params)
        batches = tr.make_batches(3, seed=7)
        sharded.comm_timing(True)
        got, want = [], []
        for i, b in enumerate(batches):
            got.append(float(tr.step(b, batches[i + 1] if i + 1 < len(batches) else None)))
            want.append(R.train_step(params, state, [t.cpu() for t in b], torch.tensor([0.7])))
        per_step = sharded.comm_timing_summary(3)
        sharded.comm_timing(False)
        assert np.allclose(got, want, atol=1e-4), (got, want)
        assert {"lookup_ids_alltoall", "lookup_rows_alltoall", "rowgrad_alltoall", "dense_grad_allreduce"} <= set(per_step)

        def close(got, want, name, noise_only=False):
            # Adam's first updates are lr * g / (|g| + eps): an element whose gradient is ~1e-4 of the typical size turns a
            # 1e-7 summation-order difference into a 1e-5 step difference (tests/test_gpu_fullsize.py has the argument):
            # every element inside the steps * lr bound, all but <= 0.2 % within 5e-6
            err = (got.cpu() - want).abs()
            assert float(err.max()) <= 2 * 3 * 1e-3 * 1.05, (name, float(err.max()))
            if not noise_only:
                assert float((err > 5e-6).float().mean()) <= 2e-3, (name, float((err > 5e-6).float().mean()), float(err.max()))

        close(tr.users.weight, params["user_id_embedding_arch.weight"], "users")
        close(tr.items.weight, params["item_id_embedding_arch.weight"], "items")
        for kk, v in tr.params.items():
            close(v, params[kk], kk, noise_only=kk in ("item_tower_arch.bias", "item_features_arch.2.bias"))
        # ... and the forced-async run is the synchronous run, bit for bit (same kernels, same order; only where the
        # collectives execute differs)
        monkeypatch.delenv("TT_COMM_FORCE_ASYNC")
        sharded.use_native_transport(None)
        tr2 = sharded.ShardedTrainer(cfg, dev, negatives="global", user_value_weights=(0.7,), dense_init=_dense_init(cfg))
        for i, b in enumerate(batches):
            tr2.step(b, batches[i + 1] if i + 1 < len(batches) else None)
        assert torch.equal(tr2.users.weight, tr.users.weight) and torch.equal(tr2.items.weight, tr.items.weight)
        for kk in tr.params:
            assert torch.equal(tr2.params[kk], tr.params[kk]), kk
    finally:
        sharded.comm_timing(False)
        sharded.use_native_transport(None)
        dist.destroy_process_group()


def test_native_comm_world1_every_collective():
    """tt_comm_* of the C ABI (csrc/comm.cpp, RCCL bound at run time) with a one-rank communicator: id, init,
    size, and every collective sharded.py uses -- at world size 1 each is the identity, which checks the binding,
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
