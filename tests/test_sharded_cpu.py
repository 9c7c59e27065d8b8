"""The multi-rank exchange logic of two_tower_models_amd.parallel on CPU: world sizes 2-4 under gloo, the four routing
kernels supplied by the torch restatement in tests/sharded_cpu_backend.py, the towers / logits / loss by the oracle's
torch expressions.  What runs from the PRODUCT is everything that decides who is sent what: `begin_lookups` /
`plan_ahead` / `routed_source` / `route_grad_rows` (through `ops.lookup_source` and `ops._route_table_grad`, the calls
the HIP autograd Functions make), `AllGatherRows` (all-gather forward, reduce-scatter backward), `ReplicatedLoss`, the
flat dense all-reduce, `shard_model_` / `full_state_dict` / `load_full_state_dict`, the row-sharded `BaselineMIPSModule`
(`sharded_topk`, `fetch_rows`).

Checked against the oracle on the CONCATENATED batch: the loss, the gradient every OWNER receives for
its row block (ids + rows, summed), the all-reduced dense gradients."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CFG = dict(n_users=53, n_items=71, D=16, F=4, B=8, H=3)
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _paths():
    for p in (ROOT, HERE, os.path.join(HERE, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _params(cfg, hist):
    g = torch.Generator().manual_seed(5)
    D, F = cfg["D"], cfg["F"]
    out = {"user_id_embedding_arch.weight": 0.5 * torch.randn(cfg["n_users"], D, generator=g),
           "item_id_embedding_arch.weight": 0.5 * torch.randn(cfg["n_items"], D, generator=g)}
    for side in ("user", "item"):
        out[f"{side}_features_arch.0.weight"] = torch.randn(256, F, generator=g) * 0.3
        out[f"{side}_features_arch.0.bias"] = torch.randn(256, generator=g) * 0.1
        out[f"{side}_features_arch.2.weight"] = torch.randn(D, 256, generator=g) * 0.06
        out[f"{side}_features_arch.2.bias"] = torch.randn(D, generator=g) * 0.1
        wide = 4 * D if (hist and side == "user") else 2 * D
        out[f"{side}_tower_arch.weight"] = torch.randn(D, wide, generator=g) * 0.2
        out[f"{side}_tower_arch.bias"] = torch.randn(D, generator=g) * 0.1
    if hist:
        for l in range(3):
            base = f"user_history_encoder.multihead_attn_layers.{l}."
            out[base + "in_proj_weight"] = torch.randn(3 * D, D, generator=g) * 0.2
            out[base + "in_proj_bias"] = torch.randn(3 * D, generator=g) * 0.05
            out[base + "out_proj.weight"] = torch.randn(D, D, generator=g) * 0.2
            out[base + "out_proj.bias"] = torch.randn(D, generator=g) * 0.05
    return out


def _batches(cfg, rank, n, seed=99):
    gen = torch.Generator().manual_seed(seed + 1000 * rank)
    B, F = cfg["B"], cfg["F"]
    return [(torch.randint(0, cfg["n_users"], (B,), generator=gen), torch.randn(B, F, generator=gen),
             torch.randint(0, cfg["n_items"], (B, cfg["H"]), generator=gen),
             torch.randint(0, cfg["n_items"], (B,), generator=gen), torch.randn(B, F, generator=gen),
             torch.randint(0, 10, (B,), generator=gen), torch.randint(0, 2, (B, 1), generator=gen).float())
            for _ in range(n)]


class _Lookup(torch.autograd.Function):
    """What every HIP Function that reads a table does (ops.EmbeddingLookup): lookup_source in the forward,
    _route_table_grad in the backward -- with the gather itself in torch."""

    @staticmethod
    def forward(ctx, weight, ids):
        from two_tower_models_amd import ops
        src, row_ids, ctx.index = ops.lookup_source(weight, ids, True)
        ctx.weight, ctx.ids = weight, ids
        return src[row_ids].view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        from two_tower_models_amd import ops
        w = ctx.weight
        return ops._route_table_grad(w, ctx.ids.reshape(-1), g.reshape(-1, w.shape[1]), ctx.index), None


def _worker(rank, world, port, outdir, cfg, hist):
    _paths()
    import torch.distributed as dist
    from oracle import cpu_ref as R
    from sharded_cpu_backend import OracleRouteKernels
    from two_tower_models_amd import parallel
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    parallel.set_route_kernels_for_tests(OracleRouteKernels())
    try:
        full = _params(cfg, hist)
        # a module with the reference's two table names: shard_model_ cuts the blocks, full_state_dict reassembles them
        holder = torch.nn.Module()
        holder.user_id_embedding_arch = torch.nn.Embedding(cfg["n_users"], cfg["D"])
        holder.item_id_embedding_arch = torch.nn.Embedding(cfg["n_items"], cfg["D"])
        holder.dense = torch.nn.ParameterDict({k.replace(".", "/"): torch.nn.Parameter(v.clone()) for k, v in full.items()
                                               if "embedding_arch" not in k})
        for name in ("user", "item"):
            w = getattr(holder, f"{name}_id_embedding_arch").weight
            w.data.copy_(full[f"{name}_id_embedding_arch.weight"])
            w._tt_is_table = True
        if rank != 0:  # replicas must come out of shard_model_ as rank 0's
            with torch.no_grad():
                for p in holder.dense.values():
                    p.add_(1.0)
        parallel.shard_model_(holder)
        back = parallel.full_state_dict(holder)
        assert all(torch.equal(back[f"{n}_id_embedding_arch.weight"], full[f"{n}_id_embedding_arch.weight"]) for n in ("user", "item"))
        assert all(torch.equal(p.data, full[k.replace("/", ".")]) for k, p in holder.dense.items()), "dense broadcast"
        parallel.load_full_state_dict(holder, back)  # round trip
        tables = {"user": holder.user_id_embedding_arch.weight, "item": holder.item_id_embedding_arch.weight}
        for w in tables.values():  # what DenseExactAdam attaches
            w._tt_rowgrads, w._tt_lookups = [], []
        dense = {k.replace("/", "."): p for k, p in holder.dense.items()}
        pe = R.positional_table(cfg["H"], cfg["D"]) if hist else None
        batches = _batches(cfg, rank, STEPS)
        if world == 2:  # every user id of step 1 on ONE owner: the most lopsided buckets there are
            batches[1][0].fill_(int(batches[1][0][0]))
        out = []
        uvw = torch.tensor([0.7])
        for s, (uid, uf, hid, iid, itf, pos, lab) in enumerate(batches):
            for w in tables.values():
                w._tt_rowgrads.clear()
                w._tt_lookups.clear()
            for p in dense.values():
                p.grad = None
            plan = {tables["user"]: [uid], tables["item"]: ([iid, hid] if hist else [iid])}
            owner_plan = parallel.begin_lookups(plan)
            # the owners' view: one local-id block per lookup, sentinel n_local where the slot is padding
            for w, blocks in owner_plan.items():
                sh = parallel.shard_of(w)
                assert len(blocks) == len(plan[w]) and all(int(b.max()) <= sh.n_local and int(b.min()) >= 0 for b in blocks)
            if s == 0:
                # batch 1 announced a step ahead, then REFILLED IN PLACE (same storage, new ids): the planned routes must
                # not be applied to it (they were counted for the old ids)
                nb = batches[1]
                parallel.plan_ahead({tables["user"]: [nb[0]], tables["item"]: ([nb[3], nb[2]] if hist else [nb[3]])})
                nb[3].copy_((nb[3] * 7 + 3) % cfg["n_items"])
            i_emb = _Lookup.apply(tables["item"], iid)
            p_all = dict(dense)
            p_item = {k: v for k, v in p_all.items()}
            f_i = R.feature_mlp(itf, p_item, "item_features_arch.")
            I = torch.cat([i_emb, f_i], dim=1) @ dense["item_tower_arch.weight"].t() + dense["item_tower_arch.bias"]
            pieces = []
            if hist:
                h_rows = _Lookup.apply(tables["item"], hid)  # [B, H, D]
                layers = R.encoder_layers_from_params(p_all)
                pieces = [R.history_encoder_forward(h_rows, layers, 4, pe).reshape(uid.shape[0], -1)]
            u_emb = _Lookup.apply(tables["user"], uid)
            f_u = R.feature_mlp(uf, p_all, "user_features_arch.")
            U = torch.cat([u_emb, f_u] + pieces, dim=1) @ dense["user_tower_arch.weight"].t() + dense["user_tower_arch.bias"]
            # global negatives + the replicated loss head on the gathered [W*B] inputs
            I_all = parallel.AllGatherRows.apply(I, "item_emb_allgather", True)
            ce = R.inbatch_rowwise_ce(U, I_all, diag_offset=rank * uid.shape[0])
            ce_g = parallel.AllGatherRows.apply(ce, "head_row_ce_allgather")
            nuv = R.normalise_value_weights(R.net_user_value(parallel.gather_no_grad(lab), uvw))
            loss = parallel.ReplicatedLoss.apply((ce_g * nuv).sum() / ce_g.shape[0])
            loss.backward()
            names = list(dense)
            flat = torch.cat([dense[k].grad.reshape(-1) for k in names])
            flat = parallel.all_reduce_dense_start(flat).wait()
            got_tab = {}
            for name, w in tables.items():
                sh = parallel.shard_of(w)
                g = torch.zeros(sh.n_local + 1, sh.dim)
                assert len(w._tt_rowgrads) == len(plan[w]) and sorted(b.index for b in w._tt_rowgrads) == list(range(len(plan[w])))
                for b in w._tt_rowgrads:
                    g.index_add_(0, b.ids, b.rows)  # sentinel rows land in the extra row
                got_tab[name] = (sh.lo, sh.hi, g[: sh.n_local].clone())
            off, got_dense = 0, {}
            for k in names:
                got_dense[k] = flat[off:off + dense[k].numel()].view_as(dense[k]).clone()
                off += dense[k].numel()
            out.append({"loss": float(loss), "tables": got_tab, "dense": got_dense, "comm": dict(parallel.comm_bytes)})
        torch.save({"steps": out, "batches": batches}, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        parallel.set_route_kernels_for_tests(None)
        dist.destroy_process_group()


@pytest.mark.parametrize("world,hist,cfg", [(2, False, CFG), (3, False, CFG), (2, True, CFG), (3, True, dict(CFG, H=5, B=6)),
                                            # fewer item rows than ranks x rows-per-rank: the last rank owns NO item row
                                            (4, False, dict(CFG, n_items=9, B=5))])
def test_exchanges_deliver_the_reference_gradients_of_the_concatenated_batch(world, hist, cfg):
    _paths()
    from oracle import cpu_ref as R
    outdir = tempfile.mkdtemp()
    mp.spawn(_worker, args=(world, _free_port(), outdir, cfg, hist), nprocs=world, join=True)
    res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    params = _params(cfg, hist)
    kw = dict(with_history=True, heads=4, pos_table=R.positional_table(cfg["H"], cfg["D"])) if hist else {}
    for s in range(STEPS):
        cat = [torch.cat([res[r]["batches"][s][k] for r in range(world)]) for k in range(7)]
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        want = R.train_forward(leaves, cat, torch.tensor([0.7]), **kw)
        grads = dict(zip(leaves, torch.autograd.grad(want, list(leaves.values()))))
        for r in range(world):
            st = res[r]["steps"][s]
            assert abs(st["loss"] - float(want)) < 1e-5, (s, r, st["loss"], float(want))
            for name, (lo, hi, g) in st["tables"].items():
                assert torch.allclose(g, grads[f"{name}_id_embedding_arch.weight"][lo:hi], atol=2e-6), (s, r, name)
            for k, g in st["dense"].items():
                assert torch.allclose(g, grads[k], atol=3e-6), (s, r, k, float((g - grads[k]).abs().max()))
            assert {"lookup_ids_alltoall", "lookup_rows_alltoall", "rowgrad_alltoall", "item_emb_allgather",
                    "dI_reduce_scatter", "dense_grad_allreduce"} <= set(st["comm"])
    # the shards tile the tables exactly
    for name, n in (("user", cfg["n_users"]), ("item", cfg["n_items"])):
        assert sum(r["steps"][0]["tables"][name][1] - r["steps"][0]["tables"][name][0] for r in res) == n


# ---------------------------------------------------------------- sharded MIPS behind BaselineMIPSModule
def _corpus(C, skewed):
    """fixture_gen's exact-arithmetic corpus; `skewed`: the first quarter of the rows doubled (exact), so that nearly every
    query's whole top-K lies in ONE block -- the case the first-try k' of parallel.sharded_topk must detect, not assume away."""
    import fixture_gen as fg
    corpus = torch.from_numpy(fg.exact_mips_corpus(C, 32))
    if skewed:
        corpus[: C // 4] *= 2.0
    return corpus


def _mips_worker(rank, world, port, outdir, C, K, bf16, how, skewed=False):
    _paths()
    import torch.distributed as dist
    import fixture_gen as fg
    from sharded_cpu_backend import OracleMipsKernels, OracleRouteKernels
    import two_tower_models_amd as A
    from two_tower_models_amd import parallel
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    parallel.set_route_kernels_for_tests(OracleRouteKernels())
    parallel.set_mips_kernels_for_tests(OracleMipsKernels())
    try:
        corpus = _corpus(C, skewed)
        _, lo, hi = parallel.block_range(C, rank, world)
        if how == "born":  # constructed under row_sharded(): the block is born on its owner, then filled
            with parallel.row_sharded():
                m = A.BaselineMIPSModule(corpus_size=C, embedding_dim=32)
            assert m.is_sharded() and m.corpus.shape[0] == hi - lo and m.corpus_size == C
            m.set_corpus(corpus[lo:hi].clone(), bf16=bf16)
        else:  # every rank holds the whole corpus, shard_corpus_ keeps its block
            m = A.BaselineMIPSModule(corpus_size=C, embedding_dim=32)
            m.corpus = corpus.clone()
            m.shard_corpus_()
            if bf16:
                m.use_bf16_storage()
        assert m.corpus_size == C and m.corpus.shape[0] == hi - lo
        assert m.corpus.dtype == (torch.bfloat16 if bf16 else torch.float32)
        q = torch.from_numpy(fg.exact_mips_queries(5 * world, 32))[rank * 5:(rank + 1) * 5]
        idx, sc, emb = m(query_embedding=q, num_items=K)  # the reference's call (ref:src/two_tower_base_retrieval.py:246-248)
        comm = dict(parallel.comm_bytes)
        idx2, sc2 = m.search(q, K)
        assert torch.equal(idx, idx2) and torch.equal(sc, sc2)
        with pytest.raises(RuntimeError, match="out of range"):
            m.search(q, C + 1)
        torch.save({"idx": idx, "sc": sc, "emb": emb, "comm": comm}, os.path.join(outdir, f"mips{rank}.pt"))
    finally:
        parallel.set_route_kernels_for_tests(None)
        parallel.set_mips_kernels_for_tests(None)
        dist.destroy_process_group()


@pytest.mark.parametrize("world,C,K,bf16,how,skewed", [(2, 700, 20, False, "born", False), (3, 100, 40, True, "cut", False),
                                                      (4, 5, 3, False, "cut", False), (2, 64, 64, True, "born", False),
                                                      # K large enough for the first-try k' (160 of 400): enough / not enough
                                                      (4, 4000, 400, False, "born", False), (4, 4000, 400, False, "cut", True)])
def test_sharded_mips_module_equals_single_device(world, C, K, bf16, how, skewed):
    """A row-sharded BaselineMIPSModule (incl. a block smaller than K, an empty block, K = the whole corpus, bf16 blocks)
    returns the unsharded module's 3-tuple: exact top-K indices + scores, and embeddings == corpus[idx]."""
    _paths()
    import fixture_gen as fg
    from oracle import cpu_ref as R
    outdir = tempfile.mkdtemp()
    mp.spawn(_mips_worker, args=(world, _free_port(), outdir, C, K, bf16, how, skewed), nprocs=world, join=True)
    corpus = _corpus(C, skewed)  # (small integers: its bf16 form is the same numbers)
    q = torch.from_numpy(fg.exact_mips_queries(5 * world, 32))
    want_idx, want_sc, want_emb = R.mips_topk(R.round_to_bf16(q) if bf16 else q, corpus, K)
    for r in range(world):
        got = torch.load(os.path.join(outdir, f"mips{r}.pt"))
        assert torch.equal(got["idx"], want_idx[r * 5:(r + 1) * 5])
        if not bf16:  # (the CPU stand-in scores fp32 queries; the HIP path's bf16 query rounding is a -m gpu matter)
            assert torch.equal(got["sc"], want_sc[r * 5:(r + 1) * 5])
        assert got["emb"].dtype == torch.float32 and torch.equal(got["emb"], want_emb[r * 5:(r + 1) * 5])
        assert {"mips_queries_allgather", "mips_lists_alltoall", "mips_rows_alltoall"} <= set(got["comm"])
        from two_tower_models_amd import parallel
        k1 = parallel.first_try_k(K, world)
        rounds = [k1] if not skewed else [k1, K]  # the skewed corpus needs the second, full-k round -- and gets it
        assert got["comm"]["mips_lists_alltoall"] == (world - 1) * 5 * 12 * sum(rounds), (got["comm"], k1)
    if K == 400:
        assert k1 == 160


# ---------------------------------------------------------------- watchdog
class _StubEvent:
    def __init__(self, done):
        self.done = done

    def query(self):
        return self.done


def test_watchdog_names_the_exchange_a_step_is_stuck_in():
    """collectives.Watchdog: steps that complete are retired; a step that does not complete within the limit produces a
    report that lists the exchanges issued since the last completed step, oldest first (stub events: no GPU needed)."""
    import time
    _paths()
    from two_tower_models_amd import collectives
    fired = []
    wd = collectives.Watchdog(seconds=0.3, on_timeout=fired.append, poll=0.05)
    try:
        collectives.note_exchange("lookup_ids_alltoall", torch.zeros(16, dtype=torch.int64))
        wd.mark(_StubEvent(True))
        time.sleep(0.2)
        assert not fired and wd._done_step == 1
        collectives.note_exchange("item_emb_allgather", torch.zeros(8, 4))
        collectives.note_exchange("dI_reduce_scatter", torch.zeros(32, 4))
        wd.mark(_StubEvent(False))  # never completes
        time.sleep(0.8)
        assert len(fired) == 1 and wd.fired
        text = fired[0]
        assert "last completed step 1" in text
        assert text.index("item_emb_allgather (128 bytes)") < text.index("dI_reduce_scatter (512 bytes)")
        assert "lookup_ids_alltoall" not in text  # issued before the step that completed
    finally:
        wd.close()
