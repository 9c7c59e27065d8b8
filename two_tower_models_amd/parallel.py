"""Row-sharded multi-GPU training BEHIND the drop-in module API (one process per GPU, RCCL over xGMI).

The reference has no parallelism of any kind; this is new design, constrained only by parity:
W ranks with B rows each compute exactly the reference's loss and update on the CONCATENATED batch of W*B rows
("global in-batch negatives").  Nothing here is a second trainer: the user still builds `TwoTowerBaseRetrieval` /
`TwoTowerWithUserHistoryEncoder` / `TwoTowerWithDebiasing` (or a subclass with its own hooks), a `DenseExactAdam` over
`model.parameters()`, and runs the reference loop (ref:train/train.py:112-125)

    loss = model.train_forward(...); optimizer.zero_grad(); loss.backward(); optimizer.step()

on every rank.  What changes is where the embedding rows live:

    with parallel.row_sharded():                  # tables created inside own only this rank's row block
        model = TwoTowerBaseRetrieval(...)
    parallel.shard_model_(model)                  # (or: build the full model anywhere, then slice it) + broadcast dense

Partitioning
  * every embedding table is split into W contiguous row blocks; a rank's `nn.Embedding.weight` IS its block
    (`weight._tt_shard` says which), its Adam moments are the optimiser's state for that Parameter, and the optimiser
    sweeps only that block (the HBM-bound part scales 1/W with no communication);
  * the batch is split by rank; dense MLP / tower / encoder / debias-head parameters are replicated.
Exchanges per step, every size known to the host before the step starts:
  1. lookups (`ops.lookup_source` -> `routed_source`): padded fixed-capacity all-to-all.  A rank's ids are bucketed by
     owner in one stable counting pass (csrc/route.hip); each owner is sent only ITS ids, `cap` slots per peer, and
     returns the rows in the same slots:   ids all_to_all [W, cap] int64,  rows all_to_all [W, cap, D].
     `cap` = the largest (requester, owner) bucket over all ranks, rounded up to 64 -- exact, no overflow path.  It is
     all-reduced (MAX) ONE STEP AHEAD from the next batch's ids (`plan_ahead`), so the host never waits for it;
     a batch that was not announced synchronises once on those few ints.  `train_forward` starts ALL of a step's
     lookups before the first tower (`begin_lookups`), so the exchanges overlap each other and the towers.
  2. item embeddings                         all_gather      [B, D] -> [W*B, D]        (`AllGatherRows`, forward)
  3. max of the value weights, loss          all_reduce      scalars                   (`GlobalWeightedMeanLoss`)
  4. partial dI over the gathered items      reduce_scatter  [W*B, D] -> [B, D]        (`AllGatherRows`, backward)
  5. embedding-row gradients: back through the lookup's slots   all_to_all [W, cap, D] (`route_grad_rows`)
  6. dense-parameter gradients, ONE flat buffer  all_reduce  ~0.5-1.5 MB               (`DenseExactAdam.step`)
Loss heads that are not the plain weighted mean (debias heads, user-overridden hooks) are evaluated REPLICATED on the
gathered [W*B]-sized head inputs with the model's own single-device code and scaled 1/W in the backward
(`ReplicatedLoss`): every cross-row term of the reference (the batch maximum, upstream's [B,1]-vs-[B] broadcast inside
the position loss) is then the reference's own expression on the concatenated batch.

The arithmetic is libtt_hotpath.so throughout; tests/ inject a CPU restatement of the four routing kernels
(`set_route_kernels_for_tests`) to exercise this file's exchange logic under gloo without a GPU.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import collectives as C


# ----------------------------------------------------------------- who owns which rows
class RowShard:
    """Contiguous row block [lo, hi) of a [n_rows, dim] table, attached to the block's Parameter as `_tt_shard`."""

    __slots__ = ("n_rows", "dim", "world", "rank", "rows_per_rank", "lo", "hi", "n_local")

    def __init__(self, n_rows: int, dim: int, world: int, rank: int):
        self.n_rows, self.dim, self.world, self.rank = int(n_rows), int(dim), int(world), int(rank)
        self.rows_per_rank, self.lo, self.hi = block_range(n_rows, rank, world)
        self.n_local = self.hi - self.lo

    def __repr__(self):
        return f"RowShard(rows [{self.lo}, {self.hi}) of {self.n_rows}, rank {self.rank}/{self.world})"


def block_range(n_rows: int, rank: int, world: int) -> Tuple[int, int, int]:
    per = (n_rows + world - 1) // world
    lo = min(rank * per, n_rows)
    return per, lo, min(lo + per, n_rows)


def shard_of(weight) -> Optional[RowShard]:
    return getattr(weight, "_tt_shard", None)


def _group() -> Tuple[int, int]:
    if not dist.is_initialized():
        raise RuntimeError("row-sharded tables need torch.distributed to be initialised (one process per GPU)")
    return dist.get_world_size(), dist.get_rank()


_BUILDING = [False]


@contextlib.contextmanager
def row_sharded():
    """Inside, the models' constructors create each embedding table as THIS rank's row block only (a 100 M-row table
    never exists in one piece).  Follow with `shard_model_(model)` (broadcasts the replicated parameters)."""
    _group()
    prev, _BUILDING[0] = _BUILDING[0], True
    try:
        yield
    finally:
        _BUILDING[0] = prev


def embedding(num_embeddings: int, embedding_dim: int) -> nn.Embedding:
    """nn.Embedding(num_embeddings, embedding_dim) (ref:src/two_tower_base_retrieval.py:70,97) -- under
    `row_sharded()` only this rank's rows of it, N(0, 1) like the whole."""
    if not _BUILDING[0]:
        return nn.Embedding(num_embeddings, embedding_dim)
    world, rank = _group()
    sh = RowShard(num_embeddings, embedding_dim, world, rank)
    emb = nn.Embedding(max(sh.n_local, 1), embedding_dim)
    with torch.no_grad():
        emb.weight.normal_(generator=block_generator(emb.weight.device, sh.lo, num_embeddings))
    emb.weight._tt_shard = sh
    return emb


def block_generator(device: torch.device, lo: int, n_rows: int) -> torch.Generator:
    """The RNG a rank draws ITS rows of a group-wide random tensor from: seeded by the process's seed AND the block's
    first row, so that ranks which all called `torch.manual_seed(s)` with the same s (common practice in distributed
    scripts) still draw different rows -- identical blocks would make rows r, r + per, r + 2 per ... of the table start
    out equal, which is not the reference's i.i.d. N(0, 1) init (ref:src/two_tower_base_retrieval.py:70,97)."""
    g = torch.Generator(device=device)
    g.manual_seed((torch.initial_seed() * 1_000_003 + 7919 * (lo + 1) + n_rows) % (2 ** 63 - 1))
    return g


def _tables(model: nn.Module) -> List[Tuple[str, nn.Parameter]]:
    return [(n, p) for n, p in model.named_parameters() if getattr(p, "_tt_is_table", False)]


@torch.no_grad()
def shard_model_(model: nn.Module, broadcast_dense: bool = True) -> nn.Module:
    """Make `model` this rank's member of a row-sharded group: every embedding table still whole is cut down to this
    rank's row block (same Parameter object, so optimisers / hooks created later see the block), and the replicated
    parameters are broadcast from rank 0 so that the replicas start bit-identical.  Call it BEFORE building the
    optimiser."""
    world, rank = _group()
    for _, p in _tables(model):
        if shard_of(p) is None:
            sh = RowShard(p.shape[0], p.shape[1], world, rank)
            block = torch.zeros(max(sh.n_local, 1), sh.dim, dtype=p.dtype, device=p.device)
            block[: sh.n_local].copy_(p.data[sh.lo:sh.hi])
            p.data = block
            p._tt_shard = sh
    if broadcast_dense and world > 1:
        dense = [p for p in model.parameters() if not getattr(p, "_tt_is_table", False)]
        flat = torch.cat([p.data.reshape(-1) for p in dense]) if dense else None
        if flat is not None:
            C.broadcast_(flat, src=0)
            off = 0
            for p in dense:
                p.data.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
    # ... and the MIPS corpus (ref:src/baseline_mips_module.py:29-30): a module still holding a whole corpus keeps its own
    # rows [lo, hi) of it -- model.forward() then searches the W blocks together (BaselineMIPSModule._sharded_*)
    mips = getattr(model, "mips_module", None)
    if mips is not None and hasattr(mips, "shard_corpus_"):
        mips.shard_corpus_()
    return model


def is_sharded(model: nn.Module) -> bool:
    return any(shard_of(p) is not None for _, p in _tables(model))


def full_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """The reference-format state_dict (whole tables under the reference's Parameter names),
    assembled on every rank: loads into the single-device modules (either implementation) unchanged."""
    out = {}
    for k, v in model.state_dict().items():
        out[k] = v.detach().clone()
    for name, p in _tables(model):
        sh = shard_of(p)
        if sh is None:
            continue
        block = p.data.new_zeros(sh.rows_per_rank, sh.dim)
        block[: sh.n_local] = p.data[: sh.n_local]
        full = C.all_gather_rows(block) if sh.world > 1 else block
        out[name] = full[: sh.n_rows].clone()
    return out


@torch.no_grad()
def load_full_state_dict(model: nn.Module, state: Dict[str, torch.Tensor]) -> None:
    """Scatter a reference-format state_dict into the row blocks / replicated parameters."""
    own = model.state_dict()
    tables = dict(_tables(model))
    for k, v in own.items():
        p = tables.get(k)
        sh = shard_of(p) if p is not None else None
        if sh is None:
            v.copy_(state[k].to(v.device))
            continue
        full = state[k]
        if tuple(full.shape) != (sh.n_rows, sh.dim):
            raise ValueError(f"{k}: expected {(sh.n_rows, sh.dim)}, got {tuple(full.shape)}")
        p.data[: sh.n_local].copy_(full[sh.lo:sh.hi].to(p.device))


# ----------------------------------------------------------------- routing kernels (csrc/route.hip, csrc/gather.hip)
class _HipRouteKernels:
    """Owner bucketing / slot assignment / owner-side localisation / row gathers on libtt_hotpath.so."""

    def __init__(self, device: torch.device):
        from . import _native
        self.N, self.lib, self.device = _native, _native.load(), device

    def route_plan(self, ids: torch.Tensor, n_rows: int, rows_per_rank: int, world: int, max_out: torch.Tensor):
        """Count this rank's ids per owner (one stable counting pass, no sort); the largest bucket is atomicMax'ed into
        the int32 scalar view `max_out`.  Ids outside [0, n_rows) raise the device-side out-of-range flag (IndexError at
        the next poll, like the single-GPU lookups)."""
        N, lib = self.N, self.lib
        n = ids.numel()
        ws = torch.empty(lib.tt_route_workspace_bytes(n, world), dtype=torch.uint8, device=self.device)
        counts = torch.empty(world, dtype=torch.int32, device=self.device)
        N.check(lib.tt_route_count(ids.data_ptr(), n, n_rows, rows_per_rank, world, counts.data_ptr(), max_out.data_ptr(),
                                   N.oob.flag(self.device).data_ptr(), ws.data_ptr(), ws.numel(), N.stream()), "tt_route_count")
        return ids, n_rows, ws, counts

    def route_build(self, planned, rows_per_rank: int, world: int, cap: int):
        N, lib = self.N, self.lib
        ids, n_rows, ws, _counts = planned
        n = ids.numel()
        send_ids = torch.empty(world * cap, dtype=torch.int64, device=self.device)
        src_of = torch.empty(world * cap, dtype=torch.int64, device=self.device)
        slot_of = torch.empty(n, dtype=torch.int64, device=self.device)
        N.check(lib.tt_route_build(ids.data_ptr(), n, n_rows, rows_per_rank, world, cap, ws.data_ptr(), ws.numel(),
                                   send_ids.data_ptr(), slot_of.data_ptr(), src_of.data_ptr(),
                                   N.oob.flag(self.device).data_ptr(), N.stream()), "tt_route_build")
        return send_ids, slot_of, src_of

    def localize(self, ids: torch.Tensor, lo: int, n_local: int) -> torch.Tensor:
        out = torch.empty_like(ids)
        self.N.check(self.lib.tt_route_localize(ids.data_ptr(), ids.numel(), lo, n_local, out.data_ptr(), self.N.stream()),
                     "tt_route_localize")
        return out

    def gather_owned(self, table: torch.Tensor, local: torch.Tensor, n_local: int) -> torch.Tensor:
        """out[i] = table[local[i]] for local[i] < n_local, a zero row for the sentinel / padding."""
        out = torch.empty(local.numel(), table.shape[1], dtype=torch.float32, device=self.device)
        if n_local <= 0:  # a rank that owns no row of this table (fewer rows than ranks): every id is the sentinel
            return out.zero_()
        N = self.N
        if table.dtype == torch.bfloat16:  # a bf16 corpus block (BaselineMIPSModule.use_bf16_storage): rows widen exactly
            N.check(self.lib.tt_gather_rows_bf16(table.data_ptr(), n_local, table.shape[1], local.data_ptr(), local.numel(),
                                                 out.data_ptr(), table.shape[1], None, N.stream()), "tt_gather_rows_bf16")
            return out
        N.check(self.lib.tt_gather_rows(table.data_ptr(), n_local, table.shape[1], local.data_ptr(), local.numel(),
                                        out.data_ptr(), table.shape[1], None, N.stream()), "tt_gather_rows")
        return out


    # ---- all lookups of a step per launch (csrc/route.hip RouteJobs): 4 launches where the per-lookup calls above took 13
    def route_plan_many(self, items, max_out: torch.Tensor):
        """items: [(ids, n_rows, rows_per_rank, world)]; max_out int32 [len(items)] is WRITTEN.  -> route_plan's tuples."""
        N, lib = self.N, self.lib
        out = []
        for c0 in range(0, len(items), N.TT_ROUTE_MAX_JOBS):
            chunk = items[c0:c0 + N.TT_ROUTE_MAX_JOBS]
            jobs = (N.RouteJob * len(chunk))()
            world = chunk[0][3]
            for k, (ids, n_rows, rpr, w) in enumerate(chunk):
                ws = torch.empty(lib.tt_route_workspace_bytes(ids.numel(), w), dtype=torch.uint8, device=self.device)
                counts = torch.empty(w, dtype=torch.int32, device=self.device)
                j = jobs[k]
                j.ids, j.n_ids, j.n_rows, j.rows_per_rank = ids.data_ptr(), ids.numel(), n_rows, rpr
                j.counts, j.max_count = counts.data_ptr(), max_out[c0 + k:c0 + k + 1].data_ptr()
                j.ws, j.ws_bytes = ws.data_ptr(), ws.numel()
                out.append((ids, n_rows, ws, counts))
            N.check(lib.tt_route_count_jobs(jobs, len(chunk), world, N.oob.flag(self.device).data_ptr(), N.stream()),
                    "tt_route_count_jobs")
        return out

    def route_build_many(self, planned, rows_per_rank, world: int, caps):
        """-> [(send_ids, slot_of, src_of)] for route_plan tuples `planned`, in one launch (the -1 fill included)."""
        N, lib = self.N, self.lib
        out = []
        for c0 in range(0, len(planned), N.TT_ROUTE_MAX_JOBS):
            chunk = planned[c0:c0 + N.TT_ROUTE_MAX_JOBS]
            jobs = (N.RouteJob * len(chunk))()
            for k, (ids, n_rows, ws, counts) in enumerate(chunk):
                cap = caps[c0 + k]
                send_ids = torch.empty(world * cap, dtype=torch.int64, device=self.device)
                src_of = torch.empty(world * cap, dtype=torch.int64, device=self.device)
                slot_of = torch.empty(ids.numel(), dtype=torch.int64, device=self.device)
                j = jobs[k]
                j.ids, j.n_ids, j.n_rows, j.rows_per_rank = ids.data_ptr(), ids.numel(), n_rows, rows_per_rank[c0 + k]
                j.counts, j.max_count, j.ws, j.ws_bytes = counts.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel()
                j.cap, j.send_ids, j.slot_of, j.src_of = cap, send_ids.data_ptr(), slot_of.data_ptr(), src_of.data_ptr()
                out.append((send_ids, slot_of, src_of))
            N.check(lib.tt_route_build_jobs(jobs, len(chunk), world, N.oob.flag(self.device).data_ptr(), N.stream()),
                    "tt_route_build_jobs")
        return out

    def serve_many(self, items):
        """Owner side.  items: [(table block, received ids, lo, n_local)] -> [(local ids, rows [n, D] fp32)], one launch."""
        N, lib = self.N, self.lib
        out = []
        for c0 in range(0, len(items), N.TT_ROUTE_MAX_JOBS):
            chunk = items[c0:c0 + N.TT_ROUTE_MAX_JOBS]
            jobs = (N.RouteServeJob * len(chunk))()
            for k, (table, ids, lo, n_local) in enumerate(chunk):
                local = torch.empty_like(ids)
                rows = torch.empty(ids.numel(), table.shape[1], dtype=torch.float32, device=self.device)
                j = jobs[k]
                j.ids, j.n_ids, j.lo, j.n_local, j.local = ids.data_ptr(), ids.numel(), lo, max(n_local, 0), local.data_ptr()
                j.table, j.dtype = table.data_ptr(), (N.TT_BF16 if table.dtype == torch.bfloat16 else N.TT_F32)
                j.dim, j.rows = table.shape[1], rows.data_ptr()
                out.append((local, rows))
            N.check(lib.tt_route_serve_jobs(jobs, len(chunk), N.stream()), "tt_route_serve_jobs")
        return out


def _plan_many(K, items, max_out):
    if hasattr(K, "route_plan_many"):
        return K.route_plan_many(items, max_out)
    max_out.zero_()  # (the per-lookup form atomicMax'es; the tests' CPU restatement takes this path)
    return [K.route_plan(ids, n_rows, rpr, w, max_out[k:k + 1]) for k, (ids, n_rows, rpr, w) in enumerate(items)]


def _build_many(K, planned, rows_per_rank, world, caps):
    if hasattr(K, "route_build_many"):
        return K.route_build_many(planned, rows_per_rank, world, caps)
    return [K.route_build(pl, rpr, world, cap) for pl, rpr, cap in zip(planned, rows_per_rank, caps)]


def _serve_many(K, items):
    if hasattr(K, "serve_many"):
        return K.serve_many(items)
    out = []
    for table, ids, lo, n_local in items:
        local = K.localize(ids, lo, n_local)
        out.append((local, K.gather_owned(table, local, n_local)))
    return out


_ROUTE_KERNELS = {}
_ROUTE_KERNELS_TEST = [None]


def set_route_kernels_for_tests(obj) -> None:
    """tests/ only: a CPU restatement of the four routing kernels, so this file's exchange logic runs under gloo on a
    box without a GPU.  The product never calls this."""
    _ROUTE_KERNELS_TEST[0] = obj


def _kernels(device: torch.device):
    if _ROUTE_KERNELS_TEST[0] is not None:
        return _ROUTE_KERNELS_TEST[0]
    if device.type != "cuda":
        raise RuntimeError("row-sharded lookups run on MI355X only: got a CPU tensor (there is no CPU path)")
    k = _ROUTE_KERNELS.get(device.index)
    if k is None:
        k = _ROUTE_KERNELS[device.index] = _HipRouteKernels(device)
    return k


# ----------------------------------------------------------------- routed lookups
class _PlannedRoutes:
    """The routing of one step's lookups as far as it can be prepared without knowing `cap`: per lookup the counted
    owner buckets, and the all-reduced bucket maxima on their way to the host."""

    __slots__ = ("key", "planned", "counts_host", "event", "keep", "ids_ref")

    def __init__(self, key, planned, counts_host, event, keep, ids_ref=()):
        self.key, self.planned, self.counts_host, self.event, self.keep = key, planned, counts_host, event, keep
        self.ids_ref = ids_ref  # the announced id tensors, alive as long as the plan (see _route_key)

    def caps(self) -> List[int]:
        if self.event is not None:
            self.event.synchronize()  # planned a step ahead: long complete, no wait
        return [max(64, (int(c) + 63) // 64 * 64) for c in self.counts_host.tolist()]


class _RoutedLookup:
    """One lookup in flight: requester side (slot_of, src_of) and owner side (local ids, sentinel n_local)."""

    __slots__ = ("shard", "n", "cap", "slot_of", "src_of", "ids_p", "rows_p", "local")

    def __init__(self, shard, n, cap, slot_of, src_of, ids_p):
        self.shard, self.n, self.cap, self.slot_of, self.src_of, self.ids_p = shard, n, cap, slot_of, src_of, ids_p
        self.rows_p, self.local = None, None


class PendingRowGrad:
    """ops.RowGrad whose rows are still travelling to their owner (all-to-all started in the backward pass); the
    optimiser's `step()` reads `.rows`, which waits for the exchange on the current stream."""

    __slots__ = ("ids", "index", "_pending", "_rows")

    def __init__(self, ids: torch.Tensor, pending, index: Optional[int]):
        self.ids, self.index, self._pending, self._rows = ids, index, pending, None

    @property
    def rows(self) -> torch.Tensor:
        if self._rows is None:
            self._rows = self._pending.wait()
            self._pending = None
        return self._rows


_planned_next: List[Optional[_PlannedRoutes]] = [None]
comm_bytes: Dict[str, int] = {}  # bytes this rank SENT to other ranks during the last step, by exchange


def _sent(tag: str, nbytes: int) -> None:
    comm_bytes[tag] = comm_bytes.get(tag, 0) + int(nbytes)


def _route_key(specs) -> tuple:
    # The announced id TENSORS themselves (a _PlannedRoutes keeps them alive in `ids_ref`, so neither the Python object
    # nor its storage can be handed to another batch while the plan exists: an address the caching allocator recycled
    # can no longer look like the announced batch) + torch's in-place version counter: a static input buffer refilled
    # with copy_() between the announcement and the step keeps its identity but not its version, and is planned again.
    # Whether a step is fed the tensors it announced is a property of the calling code, hence the same on every rank --
    # the reuse-or-replan decision (a replan contains an all-reduce) cannot differ between ranks by allocator accident.
    return tuple((id(p), id(t), t.data_ptr(), t.numel(), t._version) for p, t in specs)


def _specs(plan: Dict[nn.Parameter, Sequence[torch.Tensor]]):
    return [(p, ids) for p, blocks in plan.items() if shard_of(p) is not None for ids in blocks]


def plan_routes(specs) -> _PlannedRoutes:
    """Count each lookup's ids by owner and start the all-reduce (MAX) of the bucket maxima + its copy to the host."""
    dev = specs[0][1].device
    K = _kernels(dev)
    counts = torch.empty(len(specs), dtype=torch.int32, device=dev)
    keep, items = [], []
    for p, ids in specs:
        sh = shard_of(p)
        # a private copy: route_build ranks THESE ids against the bucket offsets counted here, whatever happens to the
        # caller's tensor in between (a writer torch does not see -- a prefetcher's raw memcpy -- would otherwise put two
        # ids into one send slot without tripping the overflow flag)
        flat = ids.reshape(-1)
        flat = flat.clone() if (flat.dtype == torch.int64 and flat.is_contiguous()) else flat.to(torch.int64).contiguous()
        keep.append(flat)
        items.append((flat, sh.n_rows, sh.rows_per_rank, sh.world))
    planned = _plan_many(K, items, counts)  # every lookup's owner histogram + scan: two launches
    if dist.get_world_size() > 1:
        C.all_reduce_start_(counts, op=dist.ReduceOp.MAX, tag="route_caps_allreduce").wait()
    if counts.is_cuda:
        host = torch.empty(len(specs), dtype=torch.int32).pin_memory()
        host.copy_(counts, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        keep.append(counts)
    else:
        host, event = counts, None
    return _PlannedRoutes(_route_key(specs), planned, host, event, keep, tuple(t for _, t in specs))


def plan_ahead(plan: Dict[nn.Parameter, Sequence[torch.Tensor]]) -> None:
    """Announce the NEXT step's lookups (`model._lookup_plan(user_id, user_history, item_id)` of the next batch): their
    routes are counted and the bucket capacity all-reduced underneath the current step, which removes the step's only
    host wait.  A scheduling hint: the plan is used only by a step that is handed the very tensors announced here,
    unmodified (identity + in-place version; the plan keeps them alive), any other batch is planned again -- results do
    not depend on it.  Every rank must call it (or none)."""
    specs = _specs(plan)
    _planned_next[0] = plan_routes(specs) if specs else None


def _start_lookups(specs, routes: _PlannedRoutes,
                   tags=("lookup_ids_alltoall", "lookup_rows_alltoall")) -> List[Tuple[nn.Parameter, _RoutedLookup]]:
    """ids to their owners, rows back: every exchange of `specs` is in flight when this returns."""
    t_ids, t_rows = tags
    dev = specs[0][1].device
    K = _kernels(dev)
    caps = routes.caps()
    lks: List[Tuple[nn.Parameter, _RoutedLookup]] = []
    shards = [shard_of(p) for p, _ in specs]
    built = _build_many(K, routes.planned, [sh.rows_per_rank for sh in shards], shards[0].world, caps)  # ONE launch
    for (p, ids), sh, cap, (send_ids, slot_of, src_of) in zip(specs, shards, caps, built):
        lks.append((p, _RoutedLookup(sh, ids.numel(), cap, slot_of, src_of,
                                     C.all_to_all_rows_start(send_ids, tag=t_ids))))
        _sent(t_ids, (sh.world - 1) * cap * 8)
    # owner side, ONE launch for every lookup: received ids -> local row numbers (sentinel n_local for padding) + their rows
    served = _serve_many(K, [(p.data, lk.ids_p.wait(), lk.shard.lo, lk.shard.n_local) for p, lk in lks])
    for (p, lk), (local, rows) in zip(lks, served):
        sh = lk.shard
        lk.local, lk.ids_p = local, None
        lk.rows_p = C.all_to_all_rows_start(rows, tag=t_rows)
        _sent(t_rows, (sh.world - 1) * lk.cap * sh.dim * 4)
    return lks


def begin_lookups(plan: Dict[nn.Parameter, Sequence[torch.Tensor]]) -> Dict[nn.Parameter, List[torch.Tensor]]:
    """Start the routed exchange of EVERY lookup of a step (tables in `plan` order, blocks in forward order) and queue
    the results on their tables for `routed_source` to hand out.  Returns the plan as the OWNERS see it -- per sharded
    table the received local-id blocks (sentinel n_local for padding and for other ranks' rows) -- which is what
    `DenseExactAdam.begin_step` plans the table step from; tables that are not sharded pass through unchanged."""
    specs = _specs(plan)
    if not specs:
        return dict(plan)
    comm_bytes.clear()
    _DEFERRED.clear()  # a backward that aborted may have left a reduce-scatter nobody waited for: never match a later tensor
    dev = specs[0][1].device
    for p, _ in specs:
        p._tt_routed = []
    routes, _planned_next[0] = _planned_next[0], None
    if routes is None or routes.key != _route_key(specs):
        routes = plan_routes(specs)
    elif routes.event is not None and dev.type == "cuda":
        torch.cuda.current_stream().wait_event(routes.event)
    out: Dict[nn.Parameter, List[torch.Tensor]] = {p: list(b) for p, b in plan.items() if shard_of(p) is None}
    for p, lk in _start_lookups(specs, routes):
        p._tt_routed.append(lk)
        out.setdefault(p, []).append(lk.local)
    return out


_LOCAL_ROWS = [False]


@contextlib.contextmanager
def local_rows():
    """Inside, lookups of a sharded table read THIS rank's block directly and the ids are local row numbers (corpus
    export: a rank runs the item tower over its own rows, no exchange)."""
    prev, _LOCAL_ROWS[0] = _LOCAL_ROWS[0], True
    try:
        yield
    finally:
        _LOCAL_ROWS[0] = prev


def routed_source(weight: nn.Parameter, ids: torch.Tensor, recording: bool):
    """ops.lookup_source for a sharded table -> (rows [W*cap, D] as received from the owners, slot of each id in it,
    lookup index).  The consuming kernel gathers by slot, so the rows are never re-ordered in HBM."""
    from . import ops
    sh = shard_of(weight)
    if _LOCAL_ROWS[0]:
        return weight, ids.reshape(-1), None
    q = getattr(weight, "_tt_routed", None)
    if q:
        lk = q.pop(0)
        if lk.n != ids.numel():
            raise RuntimeError("forward performed a lookup that differs from the one train_forward announced "
                               f"({ids.numel()} ids, announced {lk.n})")
    else:  # not announced (inference, a hook's own lookup): route it now, one host wait for the bucket capacity
        specs = [(weight, ids)]
        lk = _start_lookups(specs, plan_routes(specs))[0][1]
    rows = lk.rows_p.wait()
    idx = ops.register_lookup(weight, lk.local) if recording else None
    if recording:
        if idx is None:
            raise RuntimeError("training through a row-sharded table needs DenseExactAdam built over the model's parameters "
                               "(its table step consumes the routed row gradients); none is attached to this table")
        live = getattr(weight, "_tt_routed_live", None)
        if live is None:
            live = weight._tt_routed_live = {}
        live[idx] = lk
    return rows, lk.slot_of, idx


def route_grad_rows(weight: nn.Parameter, grad_rows: torch.Tensor, index: Optional[int]) -> None:
    """Embedding backward for a sharded table: the gradient rows go back through the lookup's slots to the owners
    (started here, waited for by the optimiser's step), where they arrive aligned with the local ids the owner served."""
    live = getattr(weight, "_tt_routed_live", None)
    lk = live.pop(index, None) if (live is not None and index is not None) else None
    if lk is None:
        raise RuntimeError("backward through a row-sharded lookup that was not recorded by a training forward")
    sh = lk.shard
    K = _kernels(grad_rows.device)
    if not grad_rows.is_contiguous():
        grad_rows = grad_rows.contiguous()
    send = K.gather_owned(grad_rows, lk.src_of, grad_rows.shape[0])  # [W*cap, D]; zero rows in the padding slots
    pend = C.all_to_all_rows_start(send, tag="rowgrad_alltoall")
    _sent("rowgrad_alltoall", (sh.world - 1) * lk.cap * sh.dim * 4)
    weight._tt_rowgrads.append(PendingRowGrad(lk.local, pend, index))
    return None


# ----------------------------------------------------------------- autograd collectives
_DEFERRED: Dict[int, "C._Pending"] = {}  # data_ptr of a reduce-scatter result that has not been waited for yet


def resolve_pending(t: Optional[torch.Tensor]) -> None:
    """Called by the backward of this package's tower Functions on their incoming gradient: if it is the result of a
    reduce-scatter that is still in flight (AllGatherRows.backward, `defer`), the current stream waits for it now."""
    if _DEFERRED and t is not None:
        p = _DEFERRED.pop(t.data_ptr(), None)
        if p is not None:
            p.wait()


_DEFER_SAFE_PRODUCERS = ("FusedTowerBackward", "LinearBackward")


class AllGatherRows(torch.autograd.Function):
    """[B, ...] per rank -> [W*B, ...] on every rank (rank order); backward: reduce-scatter (SUM) of the gradient.
    SURVEY.md 8e exchanges 3 and 5 (item embeddings / partial dI).  When the tensor gathered was produced by one of this
    package's tower Functions, the backward only STARTS the reduce-scatter: that Function's backward waits for it when
    it runs (`resolve_pending`), and whatever autograd schedules in between -- the user tower's backward -- overlaps it."""

    @staticmethod
    def forward(ctx, x, tag: str = "item_emb_allgather", defer_ok: bool = False, started=None):
        """`defer_ok`: the caller vouches that `x` has no other consumer (autograd would otherwise SUM the deferred
        gradient with the other one on the main stream, before anyone waited for the exchange).  `started`: the
        all-gather of `x` was already started (`start_all_gather`) -- whatever was queued since overlapped it."""
        ctx.tag = tag
        ctx.defer = bool(defer_ok) and x.grad_fn is not None and type(x.grad_fn).__name__ in _DEFER_SAFE_PRODUCERS
        ctx.world = dist.get_world_size()
        if ctx.world == 1 and not C._force_async():
            return x.view_as(x)
        if started is not None:
            return started.wait()
        _sent(tag, (ctx.world - 1) * x.numel() * x.element_size())
        return C.all_gather_rows_start(x.contiguous(), tag=tag).wait()

    @staticmethod
    def backward(ctx, g):
        if ctx.world == 1 and not C._force_async():
            return g, None, None, None
        back = {"item_emb_allgather": "dI_reduce_scatter"}.get(ctx.tag, ctx.tag + "_grad_reduce_scatter")
        _sent(back, (ctx.world - 1) * (g.numel() // ctx.world) * g.element_size())
        pend = C.reduce_scatter_rows_start(g.contiguous(), tag=back)
        if ctx.defer and pend.work is not None:
            _DEFERRED[pend.out.data_ptr()] = pend
            return pend.out, None, None, None
        return pend.wait(), None, None, None


def start_all_gather(x: torch.Tensor, tag: str = "item_emb_allgather"):
    """Start the all-gather of `x` now (no autograd): hand the result to `AllGatherRows.apply(x, tag, defer_ok, started)`
    once the kernels that should overlap it have been queued.  None at world size 1."""
    if dist.get_world_size() == 1 and not C._force_async():
        return None
    _sent(tag, (dist.get_world_size() - 1) * x.numel() * x.element_size())
    return C.all_gather_rows_start(x.detach().contiguous(), tag=tag)


def gather_no_grad(x: torch.Tensor, tag: str = "head_inputs_allgather") -> torch.Tensor:
    if dist.get_world_size() == 1:
        return x
    _sent(tag, (dist.get_world_size() - 1) * x.numel() * x.element_size())
    return C.all_gather_rows_start(x.contiguous(), tag=tag).wait()


class ReplicatedLoss(torch.autograd.Function):
    """A scalar every rank computed identically from gathered inputs: value unchanged, gradient scaled 1/W -- the
    gathers' reduce-scatters and the dense all-reduce (both SUMs) then add the W copies back up to exactly one."""

    @staticmethod
    def forward(ctx, loss):
        ctx.scale = 1.0 / dist.get_world_size()
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale


_ONES: Dict[tuple, torch.Tensor] = {}


def _ones(n: int, device: torch.device) -> torch.Tensor:
    key = (device.index, n)
    if key not in _ONES:
        _ONES[key] = torch.ones(n, dtype=torch.float32, device=device)
    return _ONES[key]


class GlobalWeightedMeanLoss(torch.autograd.Function):
    """ops.WeightedMeanLoss over the GLOBAL batch: mean over W*B rows of row_ce * w, w = clamp(labels . uvw, 1e-6) /
    max over ALL ranks (ref:src/two_tower_base_retrieval.py:322,334-343 on the concatenated batch).  Two launches around
    two scalar all-reduces (tt_value_weights, tt_weighted_loss_global); labels None: w = 1 (train.py's [B] labels).
    Returns the global loss on every rank; the backward hands each rank the coefficients of ITS rows."""

    @staticmethod
    def forward(ctx, row_ce, labels, uvw):
        from . import _native as N
        from . import ops
        dev = N.require_device(row_ce, labels, uvw)
        lib = N.load()
        row_ce = row_ce.contiguous()
        B = row_ce.numel()
        W = dist.get_world_size()
        if labels is not None:
            labels = ops._labels_f32(labels, "GlobalWeightedMeanLoss")
            nuv = torch.empty(B, dtype=torch.float32, device=dev)
            mx = torch.empty(1, dtype=torch.float32, device=dev)
            N.check(lib.tt_value_weights(labels.data_ptr(), B, labels.shape[1], uvw.contiguous().data_ptr(), nuv.data_ptr(),
                                         mx.data_ptr(), N.stream()), "tt_value_weights")
            if W > 1:
                C._timed_sync("scalars", C.all_reduce_, mx, op=dist.ReduceOp.MAX)
        else:
            nuv, mx = _ones(B, dev), _ones(1, dev)
        coef = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        N.check(lib.tt_weighted_loss_global(nuv.data_ptr(), mx.data_ptr(), row_ce.data_ptr(), B, float(B * W), coef.data_ptr(),
                                            loss.data_ptr(), N.stream()), "tt_weighted_loss_global")
        if W > 1:
            C._timed_sync("scalars", C.all_reduce_, loss)
            _sent("scalars", 3 * 4 * (W - 1))
        ctx.save_for_backward(coef)
        return loss

    @staticmethod
    def backward(ctx, g):
        (coef,) = ctx.saved_tensors
        return coef * g, None, None


def all_reduce_dense_start(flat: torch.Tensor):
    _sent("dense_grad_allreduce", int(2 * (dist.get_world_size() - 1) / dist.get_world_size() * flat.numel() * 4))
    return C.all_reduce_start_(flat, tag="dense_grad_allreduce")


# ----------------------------------------------------------------- sharded MIPS (BASELINE config 5)
# The implementation behind a row-sharded `BaselineMIPSModule` (baseline_mips_module.py): the module keeps the reference's
# surface -- forward(query_embedding, num_items) -> (indices, scores, embeddings), ref:src/baseline_mips_module.py:32-72 --
# and calls these two functions when its corpus is this rank's row block.
_MIPS_KERNELS_TEST = [None]


def set_mips_kernels_for_tests(obj) -> None:
    """tests/ only: an object with mips_topk / mips_merge (a CPU restatement), so the exchange logic below runs under
    gloo on a box without a GPU.  The product never calls this."""
    _MIPS_KERNELS_TEST[0] = obj


def first_try_k(k: int, world: int) -> int:
    """How many candidates a block is asked for FIRST: its expected share of the global top-k plus six standard deviations
    (k items falling into `world` equal blocks at random), rounded up to 32.  Whether that was enough is CHECKED
    (sharded_topk), never assumed."""
    share = k / world
    sd = (k * (1.0 / world) * (1.0 - 1.0 / world)) ** 0.5
    return min(k, (int(share + 6.0 * sd + 8.0) + 31) // 32 * 32)


def sharded_topk(corpus_block: torch.Tensor, row_offset: int, query: torch.Tensor, k: int, topk=None):
    """Exact top-k of every rank's own queries over a corpus whose rows are split into W contiguous blocks
    (ref:src/baseline_mips_module.py:57-61 on one block per GPU).  Every rank brings its own B queries (the same B on
    every rank); per call:
        all_gather queries                                  [B, D] -> [W*B, D]
        local exact top-k' of ALL queries on this rank's block (tt_mips_topk)
        all_to_all of the (score, global index) lists       [W, B, k'] <-> [W, B, k']   (fixed size)
        exact merge of the W*k' candidates per own query    (tt_mips_merge)
    The global top-k is a subset of the union of the per-block top-ks and every stage uses the (score desc, index asc)
    order, so with k' = k the result equals the single-device answer.  The selection / sort stages of the local search cost
    per (query, block, k'), i.e. W times the single-device work at k' = k, although a block holds only ~k / W of the answer.
    So the first round asks for k' = first_try_k(k, W) and PROVES it was enough: a block can hold an unseen member of the
    global top-k only if its LAST returned candidate sorts strictly before the merged k-th element; if that is true for any
    (query, block) on any rank (one MAX all-reduce of a flag, one host read -- this is inference), the search is repeated
    with k' = k.  `topk(q, corpus, k)`: the local search (default ops.mips_topk; the module passes its split-fp16 variant)."""
    W = dist.get_world_size()
    kern = _MIPS_KERNELS_TEST[0]
    if kern is None:
        from . import ops as kern
    if topk is None:
        topk = kern.mips_topk
    B = query.shape[0]
    comm_bytes.clear()
    q_all = C.all_gather_rows_start(query.contiguous(), tag="mips_queries_allgather").wait() if W > 1 else query
    _sent("mips_queries_allgather", (W - 1) * query.numel() * query.element_size())
    n_local = corpus_block.shape[0]
    k_try = first_try_k(k, W) if W > 1 else k
    while True:
        k_loc = min(k_try, n_local)
        if k_loc > 0:
            idx, sc = topk(q_all, corpus_block, k_loc)  # [W*B, k_loc], local row numbers
            idx = idx + row_offset
        else:  # this rank's block is empty (fewer corpus rows than ranks x rows per rank): "no candidate" only
            idx = torch.empty(q_all.shape[0], 0, dtype=torch.int64, device=q_all.device)
            sc = torch.empty(q_all.shape[0], 0, dtype=torch.float32, device=q_all.device)
        if k_loc < k_try:  # a block smaller than k': pad with "no candidate"
            pad = k_try - k_loc
            idx = torch.cat([idx, idx.new_full((idx.shape[0], pad), -1)], dim=1)
            sc = torch.cat([sc, sc.new_zeros((sc.shape[0], pad))], dim=1)
        if W == 1:
            return kern.mips_merge(sc, idx, k)
        p_idx = C.all_to_all_rows_start(idx, tag="mips_lists_alltoall")  # chunk r of the send = rank r's queries
        p_sc = C.all_to_all_rows_start(sc, tag="mips_lists_alltoall")
        _sent("mips_lists_alltoall", (W - 1) * B * k_try * 12)
        ridx, rsc = p_idx.wait().view(W, B, k_try), p_sc.wait().view(W, B, k_try)  # [source block, own query, candidate]
        out_idx, out_sc = kern.mips_merge(rsc.permute(1, 0, 2).reshape(B, W * k_try),
                                          ridx.permute(1, 0, 2).reshape(B, W * k_try), k)
        if k_try >= k:
            return out_idx, out_sc
        # was k' enough?  block b may hide a member of query q's top-k only behind a last candidate that is itself inside it
        last_i, last_s = ridx[:, :, -1], rsc[:, :, -1]  # [W, B]
        kth_i, kth_s = out_idx[:, k - 1].unsqueeze(0), out_sc[:, k - 1].unsqueeze(0)  # [1, B]
        before = (last_s > kth_s) | ((last_s == kth_s) & (last_i < kth_i)) | (kth_i < 0)
        short = ((last_i >= 0) & before).any().to(torch.int32).reshape(1)
        if not bool(C._timed_sync("mips_enough_allreduce", C.all_reduce_, short, op=dist.ReduceOp.MAX).item()):
            return out_idx, out_sc
        k_try = k  # (adversarial placement: most of a query's answer in one block) -- the plain form


class _Block:
    """What the routed-lookup machinery needs of a sharded table: `.data` (this rank's rows) and `_tt_shard`."""

    __slots__ = ("data", "_tt_shard", "_tt_routed", "__weakref__")

    def __init__(self, data: torch.Tensor, shard: RowShard):
        self.data, self._tt_shard, self._tt_routed = data, shard, []


def fetch_rows(block: torch.Tensor, shard: RowShard, idx: torch.Tensor) -> torch.Tensor:
    """block-sharded corpus[idx] -> [B, K, D] fp32 on the rank that asked (ref:src/baseline_mips_module.py:63-69): the
    global row numbers go to their owners through the lookups' own padded all-to-all (`plan_routes` / `_start_lookups`:
    ids out, rows back in the same slots), the rows are put in request order by one gather over the receive buffer.
    Inference only (no autograd).  Every rank must call it, with the same number of indices."""
    Bq, K = idx.shape
    flat = idx.reshape(-1)
    if shard.world == 1 and not C._force_async():
        return _kernels(block.device).gather_owned(block, flat, shard.n_local).view(Bq, K, shard.dim)
    tbl = _Block(block, shard)
    specs = [(tbl, flat)]
    lk = _start_lookups(specs, plan_routes(specs), tags=("mips_rows_ids_alltoall", "mips_rows_alltoall"))[0][1]
    rows = lk.rows_p.wait()  # [W*cap, D] as the owners sent them
    return _kernels(block.device).gather_owned(rows, lk.slot_of, rows.shape[0]).view(Bq, K, shard.dim)


@torch.no_grad()
def index_corpus_sharded(model: nn.Module, item_features_block: torch.Tensor, bf16: bool = False):
    """Serve what was trained: `model.index_corpus` for the usual catalogue -- item r is row r of
    the item table, `item_features_block` [hi - lo, II] the features of THIS rank's item rows [lo, hi) -- without any
    exchange (every id is this rank's own row).  Returns the model's (row-sharded) mips_module."""
    w = model.item_id_embedding_arch.weight
    sh = shard_of(w)
    if sh is None:
        raise ValueError("index_corpus_sharded: the model's item table is not row-sharded")
    if item_features_block.shape[0] != sh.n_local:
        raise ValueError(f"item_features_block: expected {sh.n_local} rows (this rank's item rows), got "
                         f"{item_features_block.shape[0]}")
    model.index_corpus(torch.arange(sh.lo, sh.hi, device=w.device), item_features_block.to(w.device, torch.float32), bf16=bf16)
    return model.mips_module


def env_world_size() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))
