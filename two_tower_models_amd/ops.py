"""Kernel wrappers and autograd Functions over libtt_hotpath.so.

Every tensor that reaches a kernel is an fp32 (or int64 id) HIP tensor owned by
torch; the C side only sees raw pointers, sizes and the current stream
(SURVEY.md 8b "Ownership").  Nothing here computes with torch ops: torch is the
allocator, the autograd tape and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
import warnings
from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as N


# ----------------------------------------------------------------- helpers
def _f32_2d(t: torch.Tensor, name: str) -> Tuple[int, int, int, int]:
    """(ptr, rows, cols, ld) of a row-major fp32 matrix view (unit column stride)."""
    if t.dtype != torch.float32 or t.dim() != 2:
        raise TypeError(f"{name}: expected a 2-D float32 tensor, got {tuple(t.shape)} {t.dtype}")
    if t.size(1) > 1 and t.stride(1) != 1:
        raise ValueError(f"{name}: columns must be contiguous")
    ld = t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0))
    return t.data_ptr(), t.size(0), t.size(1), max(ld, t.size(1))


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dim() == 2 and (t.size(1) == 1 or t.stride(1) == 1) and t.stride(0) >= t.size(1)) else t.contiguous()


def _ws(dev: torch.device, nbytes: int, slot: str = "ws") -> Tuple[Optional[int], int]:
    if nbytes <= 0:
        return None, 0
    if _aux_forks[0] and dev.type == "cuda" and torch.cuda.current_stream(dev) == N.aux_stream(dev):
        slot += "@aux"  # work forked to the third stream (AuxFork) runs NEXT TO the main stream's: its own scratch
    buf = N.scratch.get(dev, nbytes, slot)
    return buf.data_ptr(), buf.numel()


# ----------------------------------------------------------------- two independent chains, two streams
# The user tower and the item tower of a step share nothing until the logits kernel, and at the 1 M-row shapes each of
# their kernels is a few hundred small workgroups that leave most of the chip idle (tower_fwd_kernel at B = 8192: 1.1 GFLOP
# in 56 us).  AuxFork runs one of the two chains on the library's third stream: forward kernels there, and -- autograd
# runs a node's backward on the stream its forward ran on, and orders gradients that cross streams itself -- their
# backward kernels too.  Memory: a tensor allocated on one stream and read on the other is safe because every use of the
# third stream starts by waiting for the main one (here, run_on_side, the plan sorts), autograd records cross-stream
# gradients with the allocator, the chain's input tensors are recorded with it by hand (AuxFork.uses), and `.backward()`
# returns with the caller's stream ordered after both (AuxFork.joined).
# Measured (one process, alternating blocks, tools/ab_c3.py): C2 1.12 -> 1.09 ms, history model +-0, deferred P-shape step
# 1.36 -> 1.30 ms.  TT_TOWERS_SERIAL=1: one stream (A/B).
_FOLD_ASIDE = True  # (likewise: the composed-weight products of the encoder's forward on the third stream / between the layers)
_EL_WGRAD_SIDE = True  # (tools/ab_c3.py flips it: the collapsed last layer's weight half on the third stream / in line)
_CONCURRENT_TOWERS = os.environ.get("TT_TOWERS_SERIAL") is None
_FORK_MIN_ROWS = 2048  # (tests lower it: the golden batches are small)
_aux_forks = [0]  # forks whose backward pass has not ended yet (two models' steps may interleave; a forward that never gets a
                  # backward leaves it positive: bookkeeping overhead only).  0: _ws and the deferred optimiser skip their
                  # per-stream bookkeeping -- a few current-stream queries per launch, 0.15 ms per step at host-bound shapes


class AuxFork:
    """fork = AuxFork(dev) marks the point on the current stream that the forked chain depends on; `with fork:` enqueues
    the chain on the third stream and, on exit, makes the current stream wait for it -- whatever was enqueued on the
    current stream between the two runs next to the chain."""

    def __init__(self, dev: torch.device, rows: int = 1 << 30):
        # `rows`: the chain's batch size.  A fork costs the host ~0.2 ms per step (events, stream switches, autograd's
        # cross-stream bookkeeping): at the reference's default shapes (B = 256: 0.55 ms per step, host-bound) that is a
        # loss, from a few thousand rows on the kernels are long enough.  Not inside a hipGraph capture: the capture
        # works (tests), but a replayed graph with two branches is SLOWER on this runtime than the one-stream graph
        # (P-shape deferred step: 1.99 vs 1.13 ms) -- hipGraphLaunch pays for every cross-branch edge.
        self.on = _CONCURRENT_TOWERS and dev.type == "cuda" and rows >= _FORK_MIN_ROWS and not torch.cuda.is_current_stream_capturing()
        if self.on:
            # counted until the end of the backward pass this forward will get; a forward without one (validation loss under
            # no_grad) keeps the bookkeeping on for the duration of the chain only
            self.counted = torch.is_grad_enabled()
            _aux_forks[0] += 1
            self.main, self.aux = torch.cuda.current_stream(dev), N.aux_stream(dev)
            self.at = torch.cuda.Event()
            self.at.record(self.main)

    def __enter__(self):
        if self.on:
            self.aux.wait_event(self.at)
            self.ctx = torch.cuda.stream(self.aux)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
            done = torch.cuda.Event()
            done.record(self.aux)
            self.main.wait_event(done)
            if not self.counted:
                _aux_forks[0] = max(_aux_forks[0] - 1, 0)
        return False

    def uses(self, *tensors: Optional[torch.Tensor]) -> None:
        """The chain's INPUTS that were allocated on the caller's stream (the batch tensors): the chain's kernels -- the
        backward ones included, long after the caller has dropped its references -- read them on the third stream, and the
        caching allocator hands a freed block back to its own stream at once.  A training loop that moves each batch to the
        device and forgets it (everyone's) had item_features overwritten underneath the item tower's weight-gradient kernel
        once in ~1000 steps: dW1 wrong in 18 % of its elements (round 5: fuzz_train case 73, test_p_shape_train_step)."""
        if self.on:
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(self.aux)

    def joined(self, out: torch.Tensor) -> torch.Tensor:
        """The chain's result as the rest of the forward should see it: the same values; in the backward pass the first
        node of the chain, which arms the join of the third stream at the END of the backward pass.  (Autograd orders
        what it knows about -- gradients that cross streams, leaf accumulations -- but a backward function that hands row
        gradients to the optimiser's list launches nothing autograd could order the caller's stream behind.)"""
        return _JoinAuxAfterBackward.apply(out) if self.on and out.requires_grad else out


def _join_aux(dev: torch.device) -> None:
    cur, aux = torch.cuda.current_stream(dev), N.aux_stream(dev)
    if cur == aux:
        return
    if torch.cuda.is_current_stream_capturing():
        with torch.cuda.stream(aux):
            forked = torch.cuda.is_current_stream_capturing()
        if not forked:  # nothing of this capture is outstanding there (and a capturing stream may not wait for an outside event)
            return
    cur.wait_stream(aux)


class _JoinAuxAfterBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor) -> torch.Tensor:
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        dev = g.device
        def done():
            _join_aux(dev)
            _aux_forks[0] = max(_aux_forks[0] - 1, 0)

        torch.autograd.Variable._execution_engine.queue_callback(done)
        return g


# ----------------------------------------------------------------- "a fast path was not taken"
# Every tuned kernel has a correct generic form behind it; which one ran is reported once per process (a
# RuntimeWarning naming the constraint) and kept here for bench.py's `generic_paths` field -- a run at
# --embedding_dim 96 should SAY that it left most of the speed behind, not just be slow.
generic_paths: dict = {}


def note_generic(path: str, why: str) -> None:
    if path not in generic_paths:
        generic_paths[path] = why
        warnings.warn(f"two_tower_models_amd: {path} runs on the generic kernels ({why})", RuntimeWarning, stacklevel=3)


# ----------------------------------------------------------------- gradient work off the critical path
# Weight-gradient products feed nothing but the optimiser, so a backward function may hand them to the library's third
# stream (N.aux_stream: it sorts the row plans during the forward and is idle during the backward) and return at once: the
# kernels then run underneath whatever the rest of the backward pass queues on the main stream.  The main stream is made to
# wait for them ONCE, by a callback the autograd engine runs when the backward pass has finished (before `.backward()`
# returns) -- every consumer of `.grad` (this package's optimiser, torch.optim, user code) therefore sees completed
# gradients by ordinary stream order.  Operands are held until then.  Not used while a hipGraph is being captured.
# (The third stream, not a new one: ROCm multiplexes HIP streams onto a handful of hardware queues in creation order, and
# one more stream can land on the sweep's queue -- measured on the row-sharded step, round 4.)
_SIDE_GRADS = os.environ.get("TT_WGRAD_MAIN") is None  # TT_WGRAD_MAIN=1: everything in line (the safe mode under DDP-style reducers)
_side_state = {"held": [], "armed": False, "dev": None, "encoder": False, "leaves": set()}  # "encoder": the towers' weight gradients go aside too -- a HistoryEncoder forward ran since the last join, or the step is logits-bound (towers_wgrad_aside)


def towers_wgrad_aside() -> None:
    """The towers' weight gradients of the NEXT backward pass go to the third stream as well (until its join).  Called by
    the row-sharded forward in the logits-bound regime (thin row blocks, W*B negatives): after the backward logits kernel
    the main stream is a serial tail of small kernels -- slab reduce, two tower backwards, two weight-gradient products with
    their reduces, row-gradient gathers -- and, on a real node, the points where the exchanges are issued; the two
    weight-gradient products (60 us each) leave it."""
    _side_state["encoder"] = True


def _join_side_grads() -> None:
    st = _side_state
    if st["dev"] is not None:
        done = torch.cuda.Event()
        done.record(N.aux_stream(st["dev"]))
        torch.cuda.current_stream(st["dev"]).wait_event(done)
    st["held"].clear()
    st["leaves"].clear()
    st["armed"], st["dev"], st["encoder"] = False, None, False


def _deferrable(t: Optional[torch.Tensor]) -> bool:
    """May the gradient of `t` be handed to autograd before the side stream has written it?  Only if NOTHING reads it
    before the end-of-backward join: a leaf whose `.grad` is None (AccumulateGrad then steals the returned tensor without
    launching anything; a defined `.grad` means an in-place add on the main stream at once), dense contiguous layout
    (otherwise AccumulateGrad clones), and no tensor / post-accumulate hooks (hook-based reducers and clipping read the
    gradient inside the backward pass).  Hooks registered on the AccumulateGrad NODE itself (torch's DDP) cannot be seen
    from here: run with TT_WGRAD_MAIN=1 under such wrappers."""
    return (t is not None and t.is_leaf and t.grad is None and t.is_contiguous() and not t._backward_hooks
            and not getattr(t, "_post_accumulate_grad_hooks", None))


def run_on_side(dev: torch.device, fn, hold: Sequence[Optional[torch.Tensor]] = (),
                leaves: Sequence[Optional[torch.Tensor]] = ()) -> bool:
    """Run fn(True) on the third stream, after everything queued so far on the current one; -> False (and fn(False) runs in
    place, on the current stream -- `fn` picks its scratch slots by that flag: the two streams never share one) when side
    execution is off, the device is not a GPU, a graph is being captured, or no backward pass is running.
    `hold`: the INPUTS of fn (kept alive until the join) -- never its outputs: autograd's AccumulateGrad takes a returned
    gradient over as `.grad` without launching anything only if nobody else references it; a held output would be CLONED
    on the main stream, at once, before the side kernel has written it.  `leaves`: the tensors the gradients are for; the
    work is deferred only if each of them is `_deferrable` AND has not been deferred by another node of this backward pass
    (a Parameter that feeds two Functions: autograd SUMS the two gradients on the main stream as soon as both exist)."""
    if _SIDE_GRADS and dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
        fn(True)  # one stream while capturing: in line, on the side slots the eager warm-up steps have already sized
        return False
    if not (_SIDE_GRADS and dev.type == "cuda") or torch.is_grad_enabled():
        fn(False)
        return False
    st = _side_state
    for t in leaves:
        if not _deferrable(t) or id(t) in st["leaves"]:
            fn(False)
            return False
    if not st["armed"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_side_grads)
        except RuntimeError:  # not inside a backward pass: nobody would join
            fn(False)
            return False
        st["armed"], st["dev"] = True, dev
    side = N.aux_stream(dev)
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        fn(True)
    st["held"].extend(t for t in hold if t is not None)
    st["leaves"].update(id(t) for t in leaves)
    return True


def gemm(layout: int, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, M: int, Nn: int, K: int,
         bias: Optional[torch.Tensor] = None, epilogue: int = N.TT_EPI_NONE,
         aux: Optional[torch.Tensor] = None, accumulate: bool = False, slot: str = "ws") -> torch.Tensor:
    """out[M,N] (+)= op(A) op(B) (+bias) -- see tt_gemm_f32.  A/B/out may be strided row views.
    slot: which scratch buffer a split-K plan may use (calls queued on the side stream must not share the main one's)."""
    dev = N.require_device(A, B, out, bias, aux)
    lib = N.load()
    pa, _, _, lda = _f32_2d(A, "A")
    pb, _, _, ldb = _f32_2d(B, "B")
    pc, _, _, ldc = _f32_2d(out, "out")
    paux, ldaux = (None, 0)
    if aux is not None:
        paux, _, _, ldaux = _f32_2d(aux, "aux")
    wsp, wsn = _ws(dev, lib.tt_gemm_workspace_bytes(layout, M, Nn, K), slot)
    N.check(lib.tt_gemm_f32(layout, M, Nn, K, pa, lda, pb, ldb, pc, ldc, N.ptr(bias), epilogue, paux, ldaux,
                            1 if accumulate else 0, wsp, wsn, N.stream()), "tt_gemm_f32")
    return out


def gemm_tn_colsum(dy: torch.Tensor, x: torch.Tensor, dW: torch.Tensor, accumulate: bool = False,
                   db: Optional[torch.Tensor] = None, slot: str = "ws"):
    """Weight gradient and bias gradient of y = x W^T + b in one pass over dy:
    dW[N_out, K_in] (+)= dy^T x,  db[N_out] = sum over rows of dy  (tt_gemm_tn_colsum_f32)."""
    dev = N.require_device(dy, x, dW)
    lib = N.load()
    pa, Mrows, Nout, lda = _f32_2d(dy, "dy")
    pb, _, Kin, ldb = _f32_2d(x, "x")
    pc, _, _, ldc = _f32_2d(dW, "dW")
    if db is None:
        db = torch.empty(Nout, dtype=torch.float32, device=dev)
    wsp, wsn = _ws(dev, lib.tt_gemm_workspace_bytes(N.TT_GEMM_TN, Nout, Kin, Mrows), slot)
    N.check(lib.tt_gemm_tn_colsum_f32(Nout, Kin, Mrows, pa, lda, pb, ldb, pc, ldc, 1 if accumulate else 0,
                                      db.data_ptr(), wsp, wsn, N.stream()), "tt_gemm_tn_colsum_f32")
    return dW, db


def colsum(X: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    dev = N.require_device(X)
    lib = N.load()
    px, M, Nn, ldx = _f32_2d(X, "X")
    if out is None:
        out = torch.empty(Nn, dtype=torch.float32, device=dev)
    wsp, wsn = _ws(dev, lib.tt_colsum_workspace_bytes(M, Nn))
    N.check(lib.tt_colsum_f32(px, M, Nn, ldx, out.data_ptr(), wsp, wsn, N.stream()), "tt_colsum_f32")
    return out


def gather_rows_into(table: torch.Tensor, ids: torch.Tensor, out: torch.Tensor) -> None:
    """out[i, :D] = table[ids[i]] ; `out` may be a column-slice view (row stride > D)."""
    dev = N.require_device(table, ids, out)
    if ids.dtype == torch.int32:  # nn.Embedding accepts IntTensor as well
        ids = ids.to(torch.int64)
    if ids.dtype != torch.int64:
        raise TypeError("ids must be int64 or int32 (nn.Embedding's index types)")
    ids = ids.contiguous()
    pt, n_rows, D, ldt = _f32_2d(table, "table")
    if ldt != D:
        raise ValueError("embedding table must be contiguous")
    po, _, _, ldo = _f32_2d(out, "out")
    N.check(N.load().tt_gather_rows(pt, n_rows, D, ids.data_ptr(), ids.numel(), po, ldo,
                                    N.oob.flag(dev).data_ptr(), N.stream()), "tt_gather_rows")


# ----------------------------------------------------------------- row-gradient plumbing
class RowGrad:
    """One block of embedding-row gradients: `rows[i]` is dLoss/d table[ids[i]].  `index` is
    the position of the originating lookup among the table's lookups of this forward."""

    __slots__ = ("ids", "rows", "index")

    def __init__(self, ids: torch.Tensor, rows: torch.Tensor, index: Optional[int] = None):
        self.ids = ids.reshape(-1)
        self.rows = _rowmajor(rows)
        self.index = index


class RowPlan:
    """Device-side run structure of the ids looked up in a step (tt_rowgrad_plan).  Built from
    the ids alone (they are known at forward time); the gradient blocks are attached later."""

    def __init__(self, id_blocks: Sequence[torch.Tensor], n_rows: int, slot: str = "plan", defer: bool = False):
        """`defer`: only concatenate the ids (`self.ids`); the sort is launched by `build()`."""
        if len(id_blocks) > N.TT_MAX_GRAD_SOURCES:
            raise RuntimeError(f"more than {N.TT_MAX_GRAD_SOURCES} lookups of one table in a step")
        id_blocks = [b.reshape(-1) for b in id_blocks]
        dev = id_blocks[0].device
        lib = N.load()
        ids = id_blocks[0] if len(id_blocks) == 1 else torch.cat(id_blocks)
        if ids.dtype != torch.int64 or not ids.is_contiguous():
            ids = ids.to(torch.int64).contiguous()
        n = ids.numel()
        self.n, self.ids, self.n_rows, self._slot = n, ids, n_rows, slot
        self.block_sizes = [b.numel() for b in id_blocks]
        self.sorted_ids = torch.empty(n, dtype=torch.int32, device=dev)
        self.perm = torch.empty(n, dtype=torch.int32, device=dev)
        self.seg_begin = torch.empty(n + 1, dtype=torch.int32, device=dev)
        self.n_unique = torch.empty(1, dtype=torch.int32, device=dev)
        self.sources = None
        self._keep = None
        if not defer:
            self.build()

    def build(self) -> None:
        lib, dev = N.load(), self.ids.device
        wsp, wsn = _ws(dev, lib.tt_rowgrad_workspace_bytes(self.n), self._slot)
        N.check(lib.tt_rowgrad_plan(self.ids.data_ptr(), self.n, self.n_rows, self.sorted_ids.data_ptr(),
                                    self.perm.data_ptr(), self.seg_begin.data_ptr(), self.n_unique.data_ptr(),
                                    N.oob.flag(dev).data_ptr(), wsp, wsn, N.stream()), "tt_rowgrad_plan")

    @staticmethod
    def build_many(plans: Sequence["RowPlan"]) -> None:
        """build() of several deferred plans: ONE launch when every list fits the one-workgroup sort (tt_rowgrad_plan_jobs:
        a workgroup per list -- a base-model step's two sorts side by side), else one after the other."""
        plans = list(plans)
        lib = N.load()
        if 2 <= len(plans) <= N.TT_PLAN_MAX_JOBS and all(lib.tt_rowgrad_plan_jobs_supported(p.n) for p in plans):
            jobs = (N.PlanJob * len(plans))()
            for j, p in zip(jobs, plans):
                j.ids, j.n_ids, j.n_rows = p.ids.data_ptr(), p.n, p.n_rows
                j.sorted_ids, j.perm, j.seg_begin, j.n_unique = (t.data_ptr() for t in (p.sorted_ids, p.perm, p.seg_begin, p.n_unique))
            dev = plans[0].ids.device
            N.check(lib.tt_rowgrad_plan_jobs(jobs, len(plans), N.oob.flag(dev).data_ptr(), N.stream()), "tt_rowgrad_plan_jobs")
            return
        for p in plans:
            p.build()

    @classmethod
    def from_grads(cls, blocks: Sequence[RowGrad], n_rows: int) -> "RowPlan":
        plan = cls([b.ids for b in blocks], n_rows)
        plan.attach([b.rows for b in blocks])
        return plan

    def attach(self, row_blocks: Sequence[torch.Tensor]) -> None:
        """Bind the gradient rows, one block per id block, in the same order."""
        if len(row_blocks) != len(self.block_sizes):
            raise RuntimeError("number of gradient blocks differs from the number of lookups")
        src = N.GradSources()
        first = 0
        keep = []
        for k, rows in enumerate(row_blocks):
            rows = _rowmajor(rows)
            p, r, _, ld = _f32_2d(rows, "grad rows")
            if r != self.block_sizes[k]:
                raise RuntimeError("gradient block size differs from its lookup")
            src.rows[k], src.ld[k], src.first[k] = p, ld, first
            first += r
            keep.append(rows)
        src.first[len(row_blocks)] = first
        src.n_sources = len(row_blocks)
        self.sources, self._keep = src, keep


def register_lookup(weight: torch.Tensor, ids: torch.Tensor) -> Optional[int]:
    """Tell the optimiser that owns `weight` which rows this forward reads (it can then plan
    the step and start the table sweep before the gradients exist).  Returns the lookup's index."""
    reg = getattr(weight, "_tt_lookups", None)
    if reg is None or not weight.requires_grad:
        return None
    reg.append(ids.reshape(-1))
    return len(reg) - 1


class ActiveStash:
    """The optimiser's parked copy of the rows a forward is about to look up (old values): the p
    plane of the Adam side buffer, one row per lookup occurrence in announcement order
    (tt_adam_table_stash_ids).  While it is attached to a table (`weight._tt_active`), lookups
    read from here: the zero-gradient sweep may already be rewriting the table itself."""

    _positions = {}  # (concrete device index, n) -> arange(n): the same few sizes every step
    _pinned = set()  # keys whose tensor was used while a hipGraph was being captured: never evicted

    @staticmethod
    def positions_for(device: torch.device, n: int) -> torch.Tensor:
        """arange(n) on `device`, cached per CONCRETE device index (torch.device('cuda') carries None: two devices
        of one process must not share an entry) and bounded.  An entry is never created while a hipGraph is being
        captured: its contents would exist only once that graph is replayed, yet eager code would read it."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, int(n))
        pos = ActiveStash._positions.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        if pos is None:
            if capturing:
                return torch.arange(n, dtype=torch.int64, device=device)  # part of the graph, not of the cache
            if len(ActiveStash._positions) > 64:
                # evict, but never an entry a captured hipGraph has baked the address of: the graph keeps replaying reads
                # of that memory, and a cleared entry would hand it back to the allocator
                for k in [k for k in ActiveStash._positions if k not in ActiveStash._pinned]:
                    del ActiveStash._positions[k]
            pos = ActiveStash._positions[key] = torch.arange(n, dtype=torch.int64, device=device)
        elif capturing:
            ActiveStash._pinned.add(key)
        return pos

    def __init__(self, p_plane: torch.Tensor, block_sizes: Sequence[int]):
        self.p_plane = p_plane
        self.positions = ActiveStash.positions_for(p_plane.device, p_plane.shape[0])
        self.offsets = [0]
        for n in block_sizes:
            self.offsets.append(self.offsets[-1] + n)

    def slots_for(self, index: int, n: int) -> torch.Tensor:
        if index + 1 >= len(self.offsets) or self.offsets[index + 1] - self.offsets[index] != n:
            raise RuntimeError("forward performed a lookup the optimiser was not told about (begin_step mismatch)")
        return self.positions[self.offsets[index]: self.offsets[index + 1]]


class ActiveMarks:
    """Attached to a table (`weight._tt_active`) while its looked-up rows are MARKED for the sweep to step over (optim.py):
    lookups read the table itself -- the marked rows keep their old values until the finish -- but only the lookups the
    optimiser was told about: any other row may be mid-update."""

    p_plane = None

    def __init__(self, block_sizes: Sequence[int]):
        self.block_sizes = list(block_sizes)

    def slots_for(self, index: int, n: int) -> None:
        if index >= len(self.block_sizes) or self.block_sizes[index] != n:
            raise RuntimeError("forward performed a lookup the optimiser was not told about (begin_step mismatch)")


def lookup_source(weight: torch.Tensor, ids: torch.Tensor, recording: bool):
    """-> (rows_table [n, D], row_ids int64 [numel], lookup_index).  Registers the lookup with the
    table's optimiser when the forward is being recorded."""
    if getattr(weight, "_tt_shard", None) is not None:
        # row-sharded table: the rows arrive from their owners (parallel.py); the "table" the consuming kernel gathers
        # from is the receive buffer, the "ids" are the slots in it
        from . import parallel
        return parallel.routed_source(weight, ids, recording)
    lazy = getattr(weight, "_tt_lazy", None)
    if lazy is not None:  # deferred Adam: the rows must be current before anyone reads them
        lazy.catch_up(ids)
    idx = register_lookup(weight, ids) if recording else None
    act = getattr(weight, "_tt_active", None)
    if act is not None and idx is not None:
        slots = act.slots_for(idx, ids.numel())
        if act.p_plane is not None:
            return act.p_plane, slots, idx
    return weight, ids.reshape(-1), idx


def dense_grad_from_rows(blocks: Sequence[RowGrad], n_rows: int, dim: int) -> torch.Tensor:
    """The dense [n_rows, dim] embedding gradient torch.optim expects."""
    dev = blocks[0].ids.device
    plan = RowPlan.from_grads(blocks, n_rows)
    dense = torch.zeros(n_rows, dim, dtype=torch.float32, device=dev)
    N.check(N.load().tt_rowgrad_dense(C.byref(plan.sources), plan.n, dim, plan.sorted_ids.data_ptr(),
                                      plan.perm.data_ptr(), plan.seg_begin.data_ptr(),
                                      plan.n_unique.data_ptr(), dense.data_ptr(), N.stream()), "tt_rowgrad_dense")
    return dense


def _route_table_grad(weight: torch.Tensor, ids: torch.Tensor, rows: torch.Tensor,
                      index: Optional[int] = None) -> Optional[torch.Tensor]:
    """Embedding backward.  If the optimiser that owns `weight` consumes row gradients
    (two_tower_models_amd.optim.DenseExactAdam sets `_tt_rowgrads`), park them there and
    return no dense gradient; otherwise build the dense gradient torch.optim needs."""
    if getattr(weight, "_tt_shard", None) is not None and index is not None:
        from . import parallel
        return parallel.route_grad_rows(weight, rows, index)  # back through the lookup's slots to the owning ranks
    stash = getattr(weight, "_tt_rowgrads", None)
    if stash is not None:
        stash.append(RowGrad(ids, rows, index))
        return None
    return dense_grad_from_rows([RowGrad(ids, rows)], weight.shape[0], weight.shape[1])


def _resolve_pending(g: torch.Tensor) -> None:
    """Row-sharded training: an incoming tower gradient may be the result of a reduce-scatter that was only STARTED
    (parallel.AllGatherRows.backward); the current stream waits for it here, at its first use."""
    from . import parallel
    parallel.resolve_pending(g)


# ----------------------------------------------------------------- autograd Functions
_caller_grad_mode = [True]


def _recording(ctx) -> bool:
    """Is this lookup part of a forward that WILL be differentiated?  `ctx.needs_input_grad[0]` alone
    mirrors `weight.requires_grad` even under torch.no_grad(), and inside Function.forward grad mode
    is always off -- so the caller's grad mode is captured by `_LookupFunction.apply`.  An eval forward
    or index_corpus() between two training steps must not leave id blocks with the optimiser."""
    return bool(ctx.needs_input_grad[0]) and _caller_grad_mode[0]


class _LookupFunction(torch.autograd.Function):
    """Base of the Functions that read an embedding table: remembers the caller's grad mode."""

    @classmethod
    def apply(cls, *args, **kwargs):
        prev = _caller_grad_mode[0]
        _caller_grad_mode[0] = torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _caller_grad_mode[0] = prev


class EmbeddingLookup(_LookupFunction):
    """nn.Embedding.__call__ (ref:src/two_tower_base_retrieval.py:126,209)."""

    @staticmethod
    def forward(ctx, weight: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        out = torch.empty(ids.numel(), weight.shape[1], dtype=torch.float32, device=weight.device)
        src, row_ids, ctx.lookup_index = lookup_source(weight, ids, _recording(ctx))
        gather_rows_into(src, row_ids, out)
        ctx.weight = weight
        ctx.save_for_backward(ids)
        return out.view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        w = ctx.weight
        return _route_table_grad(w, ids.reshape(-1), g.reshape(-1, w.shape[1]), ctx.lookup_index), None


class Linear(torch.autograd.Function):
    """y = x W^T + b.  nn.Linear at ref:...base_retrieval.py:90-93,107-110."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = _rowmajor(x)
        M, K = x.shape
        Nn = W.shape[0]
        y = torch.empty(M, Nn, dtype=torch.float32, device=x.device)
        gemm(N.TT_GEMM_NT, x, W, y, M, Nn, K, bias=b)
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        _resolve_pending(dy)
        dy = _rowmajor(dy)
        M, K = x.shape
        Nn = W.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
            gemm(N.TT_GEMM_NN, dy, W, dx, M, K, Nn)
        dW = torch.empty(Nn, K, dtype=torch.float32, device=dy.device)
        _, db = gemm_tn_colsum(dy, x, dW)
        return dx, dW, db


class FeatureMLP(torch.autograd.Function):
    """Linear(F->256) -> ReLU -> Linear(256->D)  (ref:...base_retrieval.py:76-80)."""

    @staticmethod
    def forward(ctx, feats, W1, b1, W2, b2):
        dev = N.require_device(feats, W1, b1, W2, b2)
        feats = _rowmajor(feats)
        B, F = feats.shape
        Hd, Dm = W1.shape[0], W2.shape[0]
        h = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        gemm(N.TT_GEMM_NT, feats, W1, h, B, Hd, F, bias=b1, epilogue=N.TT_EPI_RELU)
        y = torch.empty(B, Dm, dtype=torch.float32, device=dev)
        gemm(N.TT_GEMM_NT, h, W2, y, B, Dm, Hd, bias=b2)
        ctx.save_for_backward(feats, h, W1, W2)
        return y

    @staticmethod
    def backward(ctx, dy):
        feats, h, W1, W2 = ctx.saved_tensors
        dy = _rowmajor(dy)
        dev = dy.device
        B, F = feats.shape
        Dm, Hd = W2.shape
        dW2 = torch.empty(Dm, Hd, dtype=torch.float32, device=dev)
        _, db2 = gemm_tn_colsum(dy, h, dW2)
        dh = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        gemm(N.TT_GEMM_NN, dy, W2, dh, B, Hd, Dm, epilogue=N.TT_EPI_RELU_MASK, aux=h)
        dW1 = torch.empty(Hd, F, dtype=torch.float32, device=dev)
        _, db1 = gemm_tn_colsum(dh, feats, dW1)
        dfe = None
        if ctx.needs_input_grad[0]:
            dfe = torch.empty(B, F, dtype=torch.float32, device=dev)
            gemm(N.TT_GEMM_NN, dh, W1, dfe, B, F, Hd)
        return dfe, dW1, db1, dW2, db2


class TowerInput(_LookupFunction):
    """[ table[ids] | Linear(256->D)(ReLU(Linear(F->256)(features))) ] written as two
    column slices of one [B, 2D] buffer -- the id lookup, the feature MLP and the
    torch.cat of ref:src/two_tower_base_retrieval.py:129-162 / :209-216."""

    @staticmethod
    def forward(ctx, weight, ids, feats, W1, b1, W2, b2):
        dev = N.require_device(weight, ids, feats, W1, b1, W2, b2)
        feats = _rowmajor(feats)
        B, F = feats.shape
        D = weight.shape[1]
        Dm = W2.shape[0]
        Hd = W1.shape[0]
        tin = torch.empty(B, D + Dm, dtype=torch.float32, device=dev)
        src, row_ids, ctx.lookup_index = lookup_source(weight, ids, _recording(ctx))
        gather_rows_into(src, row_ids, tin[:, :D])
        h = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        gemm(N.TT_GEMM_NT, feats, W1, h, B, Hd, F, bias=b1, epilogue=N.TT_EPI_RELU)
        gemm(N.TT_GEMM_NT, h, W2, tin[:, D:], B, Dm, Hd, bias=b2)
        ctx.weight = weight
        ctx.save_for_backward(ids, feats, h, W1, W2)
        return tin

    @staticmethod
    def backward(ctx, d_tin):
        ids, feats, h, W1, W2 = ctx.saved_tensors
        w = ctx.weight
        d_tin = _rowmajor(d_tin)
        dev = d_tin.device
        B, F = feats.shape
        D = w.shape[1]
        Dm, Hd = W2.shape
        d_f = d_tin[:, D:]
        dW2 = torch.empty(Dm, Hd, dtype=torch.float32, device=dev)
        _, db2 = gemm_tn_colsum(d_f, h, dW2)
        dh = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        gemm(N.TT_GEMM_NN, d_f, W2, dh, B, Hd, Dm, epilogue=N.TT_EPI_RELU_MASK, aux=h)
        dW1 = torch.empty(Hd, F, dtype=torch.float32, device=dev)
        _, db1 = gemm_tn_colsum(dh, feats, dW1)
        dweight = None
        if ctx.needs_input_grad[0]:
            dweight = _route_table_grad(w, ids.reshape(-1), d_tin[:, :D], ctx.lookup_index)
        return dweight, None, None, dW1, db1, dW2, db2




def fused_tower_supported(weight, feats, W1, W2, W3, extra_width: int = 0) -> bool:
    """tt_tower_fwd(_x) / tt_tower_bwd_data(_x): hidden = 256, D = d_out in {32, 64, 128}, F <= 64; a third input
    block (`extra_width` columns: the history model's [recent | mean] summary) must be 2D wide."""
    if not (weight.is_cuda and feats.dim() == 2 and feats.dtype == torch.float32):
        return False
    ok = (W3.shape[1] == 2 * weight.shape[1] + extra_width and W2.shape[0] == weight.shape[1]
          and bool(N.load().tt_tower_x_supported(weight.shape[1], feats.shape[1], W1.shape[0], W3.shape[0], extra_width)))
    if not ok:
        note_generic("tower (id lookup + feature MLP + tower Linear)",
                     f"the fused kernel takes hidden = 256, D = d_out in {{32, 64, 128}}, F <= 64, a third input block of 0 or 2D "
                     f"columns; got hidden = {W1.shape[0]}, D = {weight.shape[1]}, d_out = {W3.shape[0]}, F = {feats.shape[1]}, "
                     f"third block {extra_width}: gather + three GEMM launches per direction")
    return ok


class FusedTower(_LookupFunction):
    """One whole tower -- id lookup, feature MLP, the (never materialised) cat, tower Linear -- as one kernel per
    direction (csrc/tower.hip; ref:src/two_tower_base_retrieval.py:129-162,164-191 user, :193-219 item).
    `extra` [B, 2D]: a third block of the tower input -- the history model's [recent | mean] summary
    (ref:src/two_tower_with_user_history_encoder.py:81-83,85-122) -- with W3 [D, 4D]."""

    @staticmethod
    def forward(ctx, weight, ids, feats, W1, b1, W2, b2, W3, b3, extra=None):
        dev = N.require_device(weight, ids, feats, W1, b1, W2, b2, W3, b3)
        feats = _rowmajor(feats)
        if extra is not None:
            N.require_device(extra)
            extra = _rowmajor(extra)
            # tt_tower_*_x read the third block with 16-byte loads: an encoder that returns an offset / oddly strided view
            # (a user-supplied module may) gets a packed copy instead of TT_E_UNSUPPORTED
            if extra.data_ptr() % 16 or extra.stride(0) % 4:
                extra = extra.contiguous()
        E = 0 if extra is None else extra.shape[1]
        B, F = feats.shape
        D, Hd = weight.shape[1], W1.shape[0]
        if ids.dtype == torch.int32:
            ids = ids.to(torch.int64)
        if ids.dtype != torch.int64:
            raise TypeError("ids must be int64 or int32 (nn.Embedding's index types)")
        src, row_ids, ctx.lookup_index = lookup_source(weight, ids, _recording(ctx))
        row_ids = row_ids.contiguous()
        y = torch.empty(B, W3.shape[0], dtype=torch.float32, device=dev)
        h = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        tin = torch.empty(B, 2 * D, dtype=torch.float32, device=dev)
        W1c, W2c, W3c = W1.contiguous(), W2.contiguous(), W3.contiguous()
        N.check(N.load().tt_tower_fwd_x(src.data_ptr(), src.shape[0], row_ids.data_ptr(), feats.data_ptr(), feats.stride(0), B, D,
                                        F, Hd, W1c.data_ptr(), b1.data_ptr(), W2c.data_ptr(), b2.data_ptr(), W3c.data_ptr(),
                                        b3.data_ptr(), W3.shape[0], N.ptr(extra), extra.stride(0) if E else 0, E,
                                        y.data_ptr(), y.stride(0), h.data_ptr(), tin.data_ptr(),
                                        N.oob.flag(dev).data_ptr(), N.stream()), "tt_tower_fwd_x")
        ctx.weight = weight
        ctx.has_extra = extra is not None
        ctx.tower_leaves = (W1, b1, W2, b2, W3, b3)  # as passed in: the Parameters (ops.run_on_side: `leaves`)
        ctx.save_for_backward(ids, feats, h, tin, W2c, W3c, *(() if extra is None else (extra,)))
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, feats, h, tin, W2, W3 = ctx.saved_tensors[:6]
        extra = ctx.saved_tensors[6] if ctx.has_extra else None
        E = 0 if extra is None else extra.shape[1]
        w = ctx.weight
        dev = dy.device
        _resolve_pending(dy)
        dy = dy.contiguous()
        B, F = feats.shape
        D, Hd = w.shape[1], h.shape[1]
        d_emb = torch.empty(B, D, dtype=torch.float32, device=dev)
        d_f = torch.empty(B, D, dtype=torch.float32, device=dev)
        dh = torch.empty(B, Hd, dtype=torch.float32, device=dev)
        d_extra = torch.empty(B, E, dtype=torch.float32, device=dev) if E else None
        N.check(N.load().tt_tower_bwd_data_x(dy.data_ptr(), dy.stride(0), B, D, Hd, W2.data_ptr(), W3.data_ptr(), h.data_ptr(),
                                             d_emb.data_ptr(), D, d_f.data_ptr(), dh.data_ptr(), N.ptr(d_extra), E, E,
                                             N.stream()), "tt_tower_bwd_data_x")
        E3 = 0 if extra is None else extra.shape[1]
        outs = (torch.empty(Hd, F, dtype=torch.float32, device=dev), torch.empty(Hd, dtype=torch.float32, device=dev),
                torch.empty(D, Hd, dtype=torch.float32, device=dev), torch.empty(D, dtype=torch.float32, device=dev),
                torch.empty(D, 2 * D + E3, dtype=torch.float32, device=dev), torch.empty(D, dtype=torch.float32, device=dev))
        # (the towers' weight gradients go to the third stream only in a step that also runs the history encoder -- there
        # they are worth 0.1 ms of the 4.3 ms C3 step, underneath the encoder's backward; in the sweep-bound base model the
        # extra stream hop COSTS 0.06 ms of the 1.15 ms C2 step)
        if _side_state["encoder"]:
            run_on_side(dev, lambda on_side: tower_weight_grads(dy, tin, d_f, h, dh, feats, out=outs, extra=extra, side=on_side),
                        hold=(dy, tin, d_f, h, dh, feats, extra), leaves=ctx.tower_leaves)
        else:
            tower_weight_grads(dy, tin, d_f, h, dh, feats, out=outs, extra=extra)
        dW1, db1, dW2, db2, dW3, db3 = outs
        dweight = None
        if ctx.needs_input_grad[0]:
            dweight = _route_table_grad(w, ids.reshape(-1), d_emb, ctx.lookup_index)
        return dweight, None, None, dW1, db1, dW2, db2, dW3, db3, d_extra




class FusedTowerPair(_LookupFunction):
    """FusedTower for BOTH towers of the base model with one launch per direction (tt_tower_*_pair: blockIdx.y picks the
    tower; the kernels' bodies are FusedTower's, so the bits are too).  For steps that run on ONE stream -- a whole-step
    hipGraph, batches too small for the two-stream fork -- where one tower's 128 workgroups left half the chip idle twice in
    a row.  Inputs: the user tower's nine FusedTower arguments, then the item tower's; returns (user_emb, item_emb)."""

    @staticmethod
    def forward(ctx, *args):
        u, i = args[:9], args[9:]
        dev = N.require_device(*args)
        lib = N.load()
        sides = (N.TowerFwdSide * 2)()
        keep, saved, outs = [], [], []
        ctx.lookup_index, ctx.tower_weight = [], []
        caller_grad = _caller_grad_mode[0]
        for k, (weight, ids, feats, W1, b1, W2, b2, W3, b3) in enumerate((u, i)):
            feats = _rowmajor(feats)
            B, F = feats.shape
            D, Hd = weight.shape[1], W1.shape[0]
            if ids.dtype == torch.int32:
                ids = ids.to(torch.int64)
            if ids.dtype != torch.int64:
                raise TypeError("ids must be int64 or int32 (nn.Embedding's index types)")
            recording = bool(ctx.needs_input_grad[9 * k]) and caller_grad
            src, row_ids, idx = lookup_source(weight, ids, recording)
            row_ids = row_ids.contiguous()
            y = torch.empty(B, W3.shape[0], dtype=torch.float32, device=dev)
            h = torch.empty(B, Hd, dtype=torch.float32, device=dev)
            tin = torch.empty(B, 2 * D, dtype=torch.float32, device=dev)
            W1c, W2c, W3c = W1.contiguous(), W2.contiguous(), W3.contiguous()
            sd = sides[k]
            sd.table, sd.n_rows, sd.ids, sd.feats, sd.ldf, sd.F = src.data_ptr(), src.shape[0], row_ids.data_ptr(), feats.data_ptr(), feats.stride(0), F
            sd.W1, sd.b1, sd.W2, sd.b2, sd.W3, sd.b3 = W1c.data_ptr(), b1.data_ptr(), W2c.data_ptr(), b2.data_ptr(), W3c.data_ptr(), b3.data_ptr()
            sd.y, sd.ldy, sd.h_out, sd.tin_out = y.data_ptr(), y.stride(0), h.data_ptr(), tin.data_ptr()
            keep += [src, row_ids, feats, W1c, W2c, W3c]
            saved += [ids, feats, h, tin, W2c, W3c]
            outs.append(y)
            ctx.lookup_index.append(idx)
            ctx.tower_weight.append(weight)
        N.check(lib.tt_tower_fwd_pair(sides, B, D, Hd, N.oob.flag(dev).data_ptr(), N.stream()), "tt_tower_fwd_pair")
        ctx.save_for_backward(*saved)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, dy_u, dy_i):
        t = ctx.saved_tensors
        lib = N.load()
        dev = t[2].device
        bsides, wsides = (N.TowerBwdSide * 2)(), (N.TowerWgradSide * 2)()
        keep, per = [], []
        for k, dy in enumerate((dy_u, dy_i)):
            ids, feats, h, tin, W2, W3 = t[6 * k: 6 * k + 6]
            B, F = feats.shape
            D, Hd = tin.shape[1] // 2, h.shape[1]
            if dy is None:  # (an output nobody differentiated)
                dy = torch.zeros(B, D, dtype=torch.float32, device=dev)
            _resolve_pending(dy)
            dy = dy.contiguous()
            d_emb = torch.empty(B, D, dtype=torch.float32, device=dev)
            d_f = torch.empty(B, D, dtype=torch.float32, device=dev)
            dh = torch.empty(B, Hd, dtype=torch.float32, device=dev)
            dW1, db1 = torch.empty(Hd, F, dtype=torch.float32, device=dev), torch.empty(Hd, dtype=torch.float32, device=dev)
            dW2, db2 = torch.empty(D, Hd, dtype=torch.float32, device=dev), torch.empty(D, dtype=torch.float32, device=dev)
            dW3, db3 = torch.empty(D, 2 * D, dtype=torch.float32, device=dev), torch.empty(D, dtype=torch.float32, device=dev)
            wsp, wsn = _ws(dev, lib.tt_tower_bwd_weights_workspace_bytes(B, D, F, Hd), "tower_wgrad_pair%d" % k)
            b = bsides[k]
            b.dy, b.ldy, b.W2, b.W3, b.h = dy.data_ptr(), dy.stride(0), W2.data_ptr(), W3.data_ptr(), h.data_ptr()
            b.d_emb, b.ld_demb, b.d_f, b.dh = d_emb.data_ptr(), D, d_f.data_ptr(), dh.data_ptr()
            w = wsides[k]
            w.dy, w.ldy, w.tin, w.d_f, w.h, w.dh = dy.data_ptr(), dy.stride(0), tin.data_ptr(), d_f.data_ptr(), h.data_ptr(), dh.data_ptr()
            w.feats, w.ldf, w.F = feats.data_ptr(), feats.stride(0), F
            w.dW1, w.db1, w.dW2, w.db2, w.dW3, w.db3 = (x.data_ptr() for x in (dW1, db1, dW2, db2, dW3, db3))
            w.ws, w.ws_bytes = wsp, wsn
            keep.append(dy)
            per.append((ids, d_emb, dW1, db1, dW2, db2, dW3, db3))
        N.check(lib.tt_tower_bwd_data_pair(bsides, B, D, Hd, N.stream()), "tt_tower_bwd_data_pair")
        N.check(lib.tt_tower_bwd_weights_pair(wsides, B, D, Hd, N.stream()), "tt_tower_bwd_weights_pair")
        grads = []
        for k, (ids, d_emb, dW1, db1, dW2, db2, dW3, db3) in enumerate(per):
            dweight = None
            if ctx.needs_input_grad[9 * k]:
                dweight = _route_table_grad(ctx.tower_weight[k], ids.reshape(-1), d_emb, ctx.lookup_index[k])
            grads += [dweight, None, None, dW1, db1, dW2, db2, dW3, db3]
        return tuple(grads)


def fused_tower_pair_supported(user_args, item_args) -> bool:
    """Both towers on the tuned kernels, the same batch size, widths and hidden size, contiguous fp32 features."""
    (wu, _, fu, W1u, _, W2u, _, W3u, _), (wi, _, fi, W1i, _, W2i, _, W3i, _) = user_args, item_args
    return (wu.shape[1] == wi.shape[1] and fu.shape[0] == fi.shape[0] and W1u.shape[0] == W1i.shape[0]
            and W3u.shape[0] == W3i.shape[0] and W3u.shape[1] == 2 * wu.shape[1] and W3i.shape[1] == 2 * wi.shape[1])


def tower_weight_grads(dy, tin, d_f, h, dh, feats, out=None, extra=None, side=False):
    """(dW1, db1, dW2, db2, dW3, db3) of one tower: dW3 = dy^T [tin | extra], dW2 = d_f^T h, dW1 = dh^T feats and the
    bias sums, one product launch + one reduce (tt_tower_bwd_weights_x) instead of three tt_gemm_tn_colsum_f32 calls.
    `out`: the six tensors to write into (contiguous), else they are allocated.  `extra` [B, 2D]: the third block of
    the tower input (history model), dW3 is then [D, 4D].  `side`: the call is queued on the third stream -- it then uses
    scratch slots of its own (the main stream's products use "ws" / "tower_wgrad" concurrently)."""
    slot_fused, slot_gemm = ("tower_wgrad_side", "ws_side_t") if side else ("tower_wgrad", "ws")
    dev = dy.device
    B, D = dy.shape
    Hd, F = h.shape[1], feats.shape[1]
    E = 0 if extra is None else extra.shape[1]
    if out is not None:
        dW1, db1, dW2, db2, dW3, db3 = out
    else:
        dW3 = torch.empty(D, 2 * D + E, dtype=torch.float32, device=dev)
        dW2 = torch.empty(D, Hd, dtype=torch.float32, device=dev)
        dW1 = torch.empty(Hd, F, dtype=torch.float32, device=dev)
        db3 = torch.empty(D, dtype=torch.float32, device=dev)
        db2 = torch.empty(D, dtype=torch.float32, device=dev)
        db1 = torch.empty(Hd, dtype=torch.float32, device=dev)
    lib = N.load()
    if (lib.tt_tower_x_supported(D, F, Hd, D, E) and dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and feats.stride(1) == 1
            and all(t.is_contiguous() for t in (tin, d_f, h, dh, dW1, db1, dW2, db2, dW3, db3))
            and all(t.data_ptr() % 16 == 0 for t in (dy, tin, d_f, h, dh))
            and (extra is None or (extra.stride(1) == 1 and extra.stride(0) % 4 == 0 and extra.data_ptr() % 16 == 0))):
        wsp, wsn = _ws(dev, lib.tt_tower_bwd_weights_x_workspace_bytes(B, D, F, Hd, E), slot_fused)
        N.check(lib.tt_tower_bwd_weights_x(dy.data_ptr(), dy.stride(0), tin.data_ptr(), d_f.data_ptr(), h.data_ptr(), dh.data_ptr(),
                                           feats.data_ptr(), feats.stride(0), N.ptr(extra), extra.stride(0) if E else 0, E,
                                           B, D, F, Hd, dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), db2.data_ptr(),
                                           dW3.data_ptr(), db3.data_ptr(), wsp, wsn, N.stream()), "tt_tower_bwd_weights_x")
        return dW1, db1, dW2, db2, dW3, db3
    if extra is not None:
        gemm_tn_colsum(dy, tin, dW3[:, :2 * D], db=db3, slot=slot_gemm)
        gemm(N.TT_GEMM_TN, dy, extra, dW3[:, 2 * D:], D, E, B, slot=slot_gemm)
        gemm_tn_colsum(d_f, h, dW2, db=db2, slot=slot_gemm)
        gemm_tn_colsum(dh, feats, dW1, db=db1, slot=slot_gemm)
        return dW1, db1, dW2, db2, dW3, db3
    gemm_tn_colsum(dy, tin, dW3, db=db3, slot=slot_gemm)
    gemm_tn_colsum(d_f, h, dW2, db=db2, slot=slot_gemm)
    gemm_tn_colsum(dh, feats, dW1, db=db1, slot=slot_gemm)
    return dW1, db1, dW2, db2, dW3, db3


# TT_ENC_GENERIC=1 (tests): the history encoder's plain composition -- every layer in full (in-projection, attention,
# out-projection; the last one's out-projection for row 0 only) -- instead of the algebraic shortcuts below; it is
# also what shapes outside the shortcut kernels' limits run, so it has to stay correct
_ENC_GENERIC = os.environ.get("TT_ENC_GENERIC") is not None


def kept_logits_supported(U: torch.Tensor, I: torch.Tensor) -> bool:
    """tt_inbatch_ce_fwd_du_keep / tt_inbatch_ce_bwd_kept: D in {32, 64, 128}, 16-B aligned rows."""
    D = U.shape[1]
    return (D in (32, 64, 128) and I.shape[0] < 4 * 1024 * 1024 - 256 and U.stride(0) % 4 == 0 and I.stride(0) % 4 == 0
            and U.data_ptr() % 16 == 0 and I.data_ptr() % 16 == 0 and os.environ.get("TT_CE_NO_DMA") is None)


# optimisers whose table sweep waits for the backward logits kernel to be queued (DenseExactAdam._begin_overlapped, hold_sweep)
held_sweeps = weakref.WeakSet()


def _release_held_sweeps(before_kernel: Optional[torch.cuda.Event]) -> None:
    for opt in list(held_sweeps):
        opt.release_sweep(after=before_kernel)


class InBatchSoftmaxCE(torch.autograd.Function):
    """row_ce[i] = logsumexp_j (U I^T)[i, j] - (U I^T)[i, i + diag_offset]
    (torch.matmul + F.cross_entropy(reduction="none"), ref:...base_retrieval.py:287-312)."""

    @staticmethod
    def forward(ctx, U, I, diag_offset: int = 0, keep_logits: Optional[bool] = None):
        """`keep_logits`: write the [M, N] logits out in the forward so the item-side backward does not
        recompute them (3 instead of 4 logit-sized products, for M*N*4 B of HBM each way).  Default:
        only for wide negative sets (N >= 4 M, i.e. several ranks' items per user) -- at N = M the
        step is bound by the Adam sweep's HBM traffic and the extra bytes cost more than the MFMAs."""
        dev = N.require_device(U, I)
        U, I = _rowmajor(U), _rowmajor(I)
        M, D = U.shape
        Nn = I.shape[0]
        lib = N.load()
        lse = torch.empty(M, dtype=torch.float32, device=dev)
        ce = torch.empty(M, dtype=torch.float32, device=dev)
        wsp, wsn = _ws(dev, lib.tt_inbatch_ce_workspace_bytes(M, Nn, D))
        pu, _, _, ldu = _f32_2d(U, "U")
        pi, _, _, ldi = _f32_2d(I, "I")
        ctx.diag_offset = diag_offset
        ctx.kept = ctx.kept16 = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ldu == D and ldi == D and ce16_usable(U, I) \
                and 0 <= diag_offset <= Nn - M:
            du_unit = torch.empty(M, D, dtype=torch.float32, device=dev)
            w16p, w16n = _ws(dev, lib.tt_ce16_workspace_bytes(M, Nn, D), "ce16")
            # no logits buffer unless asked for (_CE16_KEEP, the pair's first form): the backward forms the tiles again
            ctx.kept16 = torch.empty(M * Nn, dtype=torch.float32, device=dev) if _CE16_KEEP else True
            N.check(lib.tt_ce16_fwd_du_keep(pu, D, pi, D, M, Nn, D, diag_offset, lse.data_ptr(), ce.data_ptr(), du_unit.data_ptr(), D,
                                            ctx.kept16.data_ptr() if _CE16_KEEP else None, M * Nn * 4 if _CE16_KEEP else 0, w16p, w16n,
                                            N.stream()), "tt_ce16_fwd_du_keep")
            ctx.save_for_backward(U, I, lse, du_unit)
            return ce
        if ctx.needs_input_grad[0]:
            # training: the forward also accumulates E[i] = sum_j p_ij I_j, which IS the user-side
            # gradient up to the row factor -- the backward then only runs the item-side kernel
            du_unit = torch.empty(M, D, dtype=torch.float32, device=dev)
            if D > 128:
                note_generic("in-batch softmax CE", f"D = {D} > 128: logits materialised per row chunk + library GEMMs "
                                                    "(csrc/ce_wide.hip) instead of the register-stationary kernels")
            if keep_logits is None:
                keep_logits = Nn >= 4 * M
            if keep_logits and ctx.needs_input_grad[1] and not kept_logits_supported(U, I):
                note_generic("in-batch softmax CE backward (wide negative sets)",
                             f"kept logits need D in {{32, 64, 128}}, N < 4 Mi and 16-B aligned rows; got D = {D}, N = {Nn}: "
                             "the item-side backward recomputes the logits (4 instead of 3 logit-sized products)")
            keep_logits = bool(keep_logits) and ctx.needs_input_grad[1] and kept_logits_supported(U, I)
            if keep_logits:
                zn = lib.tt_inbatch_ce_logits_bytes(M, Nn)
                ctx.kept = torch.empty(zn, dtype=torch.uint8, device=dev)
                N.check(lib.tt_inbatch_ce_fwd_du_keep(pu, ldu, pi, ldi, M, Nn, D, diag_offset, lse.data_ptr(),
                                                      ce.data_ptr(), du_unit.data_ptr(), D, ctx.kept.data_ptr(), zn,
                                                      wsp, wsn, N.stream()), "tt_inbatch_ce_fwd_du_keep")
            else:
                N.check(lib.tt_inbatch_ce_fwd_du(pu, ldu, pi, ldi, M, Nn, D, diag_offset, lse.data_ptr(), ce.data_ptr(),
                                                 du_unit.data_ptr(), D, wsp, wsn, N.stream()), "tt_inbatch_ce_fwd_du")
            ctx.save_for_backward(U, I, lse, du_unit)
            return ce
        N.check(lib.tt_inbatch_ce_fwd(pu, ldu, pi, ldi, M, Nn, D, diag_offset, lse.data_ptr(), ce.data_ptr(),
                                      wsp, wsn, N.stream()), "tt_inbatch_ce_fwd")
        ctx.save_for_backward(U, I, lse)
        return ce

    @staticmethod
    def backward(ctx, d_ce):
        saved = ctx.saved_tensors
        U, I, lse = saved[:3]
        du_unit = saved[3] if len(saved) > 3 else None
        dev = U.device
        M, D = U.shape
        Nn = I.shape[0]
        lib = N.load()
        coef = d_ce.contiguous()
        dU = torch.empty(M, D, dtype=torch.float32, device=dev)
        if du_unit is not None:
            N.check(lib.tt_scale_rows(du_unit.data_ptr(), D, coef.data_ptr(), M, D, dU.data_ptr(), D, N.stream()), "tt_scale_rows")
        dI = torch.empty(Nn, D, dtype=torch.float32, device=dev)
        wsp, wsn = _ws(dev, lib.tt_inbatch_ce_workspace_bytes(M, Nn, D))
        pu, _, _, ldu = _f32_2d(U, "U")
        pi, _, _, ldi = _f32_2d(I, "I")
        before = None
        if held_sweeps:  # a table sweep waits for this kernel to be in the queue: see DenseExactAdam._begin_overlapped
            before = torch.cuda.Event()
            before.record()
        try:
            return InBatchSoftmaxCE._item_side(ctx, lib, dev, M, Nn, D, pu, ldu, pi, ldi, lse, coef, du_unit, dU, dI, wsp, wsn)
        finally:
            if before is not None:
                _release_held_sweeps(before)

    @staticmethod
    def _item_side(ctx, lib, dev, M, Nn, D, pu, ldu, pi, ldi, lse, coef, du_unit, dU, dI, wsp, wsn):
        if ctx.kept16 is not None:  # the split-fp16 pair's backward
            w16p, w16n = _ws(dev, lib.tt_ce16_workspace_bytes(M, Nn, D), "ce16")
            if ctx.kept16 is True:  # (images formed again: other products may have used the workspace slot since the forward)
                N.check(lib.tt_ce16_bwd_recompute(pu, D, pi, D, M, Nn, D, ctx.diag_offset, lse.data_ptr(), coef.data_ptr(), dI.data_ptr(), D,
                                                  w16p, w16n, 0, N.stream()), "tt_ce16_bwd_recompute")
            else:
                N.check(lib.tt_ce16_bwd_kept(pu, D, M, Nn, D, ctx.diag_offset, lse.data_ptr(), coef.data_ptr(), ctx.kept16.data_ptr(),
                                             M * Nn * 4, dI.data_ptr(), D, w16p, w16n, N.stream()), "tt_ce16_bwd_kept")
            ctx.kept16 = None
            return dU, dI, None, None
        if ctx.kept is not None:  # item side from the logits the forward kept
            N.check(lib.tt_inbatch_ce_bwd_kept(pu, ldu, M, Nn, D, ctx.diag_offset, lse.data_ptr(), coef.data_ptr(),
                                               ctx.kept.data_ptr(), ctx.kept.numel(), dI.data_ptr(), D, wsp, wsn,
                                               N.stream()), "tt_inbatch_ce_bwd_kept")
            ctx.kept = None
            return dU, dI, None, None
        N.check(lib.tt_inbatch_ce_bwd(pu, ldu, pi, ldi, M, Nn, D, ctx.diag_offset, lse.data_ptr(),
                                      coef.data_ptr(), None if du_unit is not None else dU.data_ptr(), D,
                                      dI.data_ptr(), D, wsp, wsn, N.stream()), "tt_inbatch_ce_bwd")
        return dU, dI, None, None


def _labels_f32(labels: torch.Tensor, what: str) -> torch.Tensor:
    """The kernels read `const float*`.  Integer / bool labels promote to float32 in the reference's
    `labels * user_value_weights` too, so casting them is exact; any other float width would change
    the reference's result dtype and is refused (the models route those to the torch expressions)."""
    if labels.dtype == torch.float32:
        return labels.contiguous()
    if labels.dtype in (torch.bool, torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
        return labels.to(torch.float32).contiguous()
    raise TypeError(f"{what}: labels must be float32 or an integer / bool type, got {labels.dtype}")


def labels_fusable(labels: torch.Tensor) -> bool:
    return labels.dtype in (torch.float32, torch.bool, torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64)


class WeightedMeanLoss(torch.autograd.Function):
    """mean_i(row_ce[i] * w[i]) with w = clamp(labels @ uvw, 1e-6) / max(...)
    (ref:...base_retrieval.py:322,334-343, 2-D labels, identity debias hook).  labels None: w = 1,
    the value the same expressions produce for train.py's 1-D [B] labels."""

    @staticmethod
    def forward(ctx, row_ce, labels, uvw):
        dev = N.require_device(row_ce, labels, uvw)
        if uvw.dtype != torch.float32 or row_ce.dtype != torch.float32:
            raise TypeError("WeightedMeanLoss: row_ce and user_value_weights must be float32")
        row_ce, uvw = row_ce.contiguous(), uvw.contiguous()
        if labels is not None:
            labels = _labels_f32(labels, "WeightedMeanLoss")
            B, T = labels.shape
            if uvw.numel() != T or row_ce.numel() != B:
                raise RuntimeError("WeightedMeanLoss: labels [B, T], user_value_weights [T], row_ce [B]")
        else:
            B, T = row_ce.numel(), 1
        w = torch.empty(B, dtype=torch.float32, device=dev)
        coef = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        N.check(N.load().tt_weighted_mean_loss(N.ptr(labels), B, T, uvw.data_ptr(), row_ce.data_ptr(),
                                               w.data_ptr(), coef.data_ptr(), loss.data_ptr(), N.stream()),
                "tt_weighted_mean_loss")
        ctx.save_for_backward(coef)
        return loss

    @staticmethod
    def backward(ctx, g):
        (coef,) = ctx.saved_tensors
        return coef * g, None, None


# EXPLORATORY: InBatchSoftmaxCE through the split-fp16 pair (csrc/ce_f16x2.hip) where its shapes allow
# (D = 128, M % 256 == 0, N % 1024 == 0, contiguous rows) -- fp32-grade results on the fp16 matrix pipe.  Never the default.
_CE16_KEEP = False  # A/B (tests flip it): the split-fp16 pair with kept logits (its first form) instead of recomputed ones
_CE_F16X2 = os.environ.get("TT_CE_F16X2") is not None


def ce16_usable(U: torch.Tensor, I: torch.Tensor) -> bool:
    return bool(_CE_F16X2 and U.is_cuda and U.dim() == 2 and I.dim() == 2 and U.dtype == torch.float32 and I.dtype == torch.float32
                and N.load().tt_ce16_supported(U.shape[0], I.shape[0], U.shape[1]))


def fused_loss_supported(U: torch.Tensor, I: torch.Tensor, labels: Optional[torch.Tensor], uvw: torch.Tensor) -> bool:
    """InBatchSoftmaxWeightedLoss: a training forward (U needs a gradient) with in-batch negatives only (N < 4 M: the wide
    form keeps its logits and has its own forward), float32 everywhere, one label row per user row."""
    if ce16_usable(U, I):
        return False  # the split-fp16 pair is a two-op path: InBatchSoftmaxCE + WeightedMeanLoss
    return bool(U.is_cuda and U.requires_grad and torch.is_grad_enabled() and U.dim() == 2 and I.dim() == 2
                and U.dtype == torch.float32 and I.dtype == torch.float32 and uvw.dtype == torch.float32
                and I.shape[0] < 4 * U.shape[0]
                and (labels is None or (labels.dim() == 2 and labels.shape[0] == U.shape[0] and labels.shape[1] == uvw.numel()
                                        and labels_fusable(labels))))


class InBatchSoftmaxWeightedLoss(torch.autograd.Function):
    """WeightedMeanLoss(InBatchSoftmaxCE(U, I), labels, uvw) as one op: the loss head runs in the launch that finishes
    the forward (tt_inbatch_ce_fwd_du_loss; ref:src/two_tower_base_retrieval.py:287-312,322,334-343), and the backward
    starts with ONE launch that forms dL/dce * g and dU (tt_scale_rows_g) instead of an elementwise multiply and a row
    scaling.  Same values as the two ops, bit for bit; two launches fewer on the step's critical path."""

    @staticmethod
    def forward(ctx, U, I, labels, uvw):
        dev = N.require_device(U, I, labels, uvw)
        U, I, uvw = _rowmajor(U), _rowmajor(I), uvw.contiguous()
        M, D = U.shape
        Nn = I.shape[0]
        T = 1
        if labels is not None:
            labels = _labels_f32(labels, "InBatchSoftmaxWeightedLoss")
            T = labels.shape[1]
        if D > 128:
            note_generic("in-batch softmax CE", f"D = {D} > 128: logits materialised per row chunk + library GEMMs "
                                                "(csrc/ce_wide.hip) instead of the register-stationary kernels")
        lib = N.load()
        lse = torch.empty(M, dtype=torch.float32, device=dev)
        ce = torch.empty(M, dtype=torch.float32, device=dev)
        w = torch.empty(M, dtype=torch.float32, device=dev)
        coef = torch.empty(M, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        du_unit = torch.empty(M, D, dtype=torch.float32, device=dev)
        wsp, wsn = _ws(dev, lib.tt_inbatch_ce_workspace_bytes(M, Nn, D))
        pu, _, _, ldu = _f32_2d(U, "U")
        pi, _, _, ldi = _f32_2d(I, "I")
        N.check(lib.tt_inbatch_ce_fwd_du_loss(pu, ldu, pi, ldi, M, Nn, D, 0, N.ptr(labels), T, uvw.data_ptr(), lse.data_ptr(),
                                              ce.data_ptr(), du_unit.data_ptr(), D, w.data_ptr(), coef.data_ptr(), loss.data_ptr(),
                                              wsp, wsn, N.stream()), "tt_inbatch_ce_fwd_du_loss")
        ctx.save_for_backward(U, I, lse, du_unit, coef)
        return loss

    @staticmethod
    def backward(ctx, g):
        U, I, lse, du_unit, coef = ctx.saved_tensors
        dev = U.device
        M, D = U.shape
        Nn = I.shape[0]
        lib = N.load()
        g = g.contiguous()
        dU = torch.empty(M, D, dtype=torch.float32, device=dev)
        coef_g = torch.empty(M, dtype=torch.float32, device=dev)
        N.check(lib.tt_scale_rows_g(du_unit.data_ptr(), D, coef.data_ptr(), g.data_ptr(), M, D, dU.data_ptr(), D, coef_g.data_ptr(),
                                    N.stream()), "tt_scale_rows_g")
        dI = torch.empty(Nn, D, dtype=torch.float32, device=dev)
        wsp, wsn = _ws(dev, lib.tt_inbatch_ce_workspace_bytes(M, Nn, D))
        pu, _, _, ldu = _f32_2d(U, "U")
        pi, _, _, ldi = _f32_2d(I, "I")
        N.check(lib.tt_inbatch_ce_bwd(pu, ldu, pi, ldi, M, Nn, D, 0, lse.data_ptr(), coef_g.data_ptr(), None, D,
                                      dI.data_ptr(), D, wsp, wsn, N.stream()), "tt_inbatch_ce_bwd")
        return dU, dI, None, None


class DebiasedWeightedLoss(torch.autograd.Function):
    """mean_i(row_ce_i * w_i) + aux of a debias head (ref:src/two_tower_with_debiasing.py:77-129,
    ref:src/two_tower_with_position_debiased_weights.py:76-113, ref:src/two_tower_with_user_debiased_weights.py:100-135,
    each on ref:src/two_tower_base_retrieval.py:322-345), in two kernels forward and three backward
    (tt_debias_loss_fwd / _bwd).  labels [B, T], position [B] int64, user_embedding [B, DI].
    mode N.TT_DEBIAS_COMBINED: pos_table [n_pos, 1], lin_w [1, DI + 1], lin_b [1];  TT_DEBIAS_POSITION: pos_table only
    (lin_w = lin_b = None);  TT_DEBIAS_USER: lin_w [1, DI], lin_b [1] (pos_table = None, position ignored)."""

    @staticmethod
    def forward(ctx, row_ce, labels, uvw, position, user_embedding, pos_table, lin_w, lin_b, mode=0):
        dev = N.require_device(row_ce, labels, uvw, position, user_embedding, pos_table, lin_w, lin_b)
        lib = N.load()
        labels, position = _labels_f32(labels, "DebiasedWeightedLoss"), position.contiguous()
        if any(t is not None and t.dtype != torch.float32 for t in (row_ce, uvw, user_embedding, pos_table, lin_w, lin_b)):
            raise TypeError("DebiasedWeightedLoss: float32 operands expected")
        row_ce, ue = row_ce.contiguous(), _rowmajor(user_embedding)
        pos_table = pos_table.contiguous() if pos_table is not None else None
        lin_w = lin_w.contiguous() if lin_w is not None else None
        lin_b = lin_b.contiguous() if lin_b is not None else None
        B, T = labels.shape
        pue, _, DI, ld_ue = _f32_2d(ue, "user_embedding")
        want_w = {N.TT_DEBIAS_COMBINED: DI + 1, N.TT_DEBIAS_POSITION: None, N.TT_DEBIAS_USER: DI}[mode]
        if (position.dtype != torch.int64 or uvw.numel() != T
                or (mode != N.TT_DEBIAS_USER and (pos_table is None or pos_table.shape[1] != 1))
                or (want_w is not None and (lin_w is None or lin_b is None or lin_w.numel() != want_w))):
            raise TypeError("DebiasedWeightedLoss: position int64 [B], pos_table [n_pos, 1], lin_w [1, DI + 1] (combined) / "
                            "[1, DI] (user-only), uvw [T]")
        n_pos = pos_table.shape[0] if pos_table is not None else 1
        wsn = lib.tt_debias_loss_workspace_bytes(B, DI, n_pos)
        ws = torch.empty(wsn, dtype=torch.uint8, device=dev)  # kept for the backward (n, p, e, r, scalars)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        N.check(lib.tt_debias_loss_fwd(mode, row_ce.data_ptr(), labels.data_ptr(), B, T, uvw.data_ptr(), position.data_ptr(),
                                       n_pos, N.ptr(pos_table), pue, ld_ue, DI, N.ptr(lin_w), N.ptr(lin_b), loss.data_ptr(),
                                       ws.data_ptr(), wsn, N.oob.flag(dev).data_ptr(), N.stream()), "tt_debias_loss_fwd")
        ctx.mode, ctx.n_pos = mode, n_pos
        ctx.have = (pos_table is not None, lin_w is not None)
        ctx.save_for_backward(row_ce, position, ue, ws, *(t for t in (pos_table, lin_w, lin_b) if t is not None))
        return loss

    @staticmethod
    def backward(ctx, g):
        row_ce, position, ue, ws, *rest = ctx.saved_tensors
        has_pos, has_lin = ctx.have
        pos_table = rest.pop(0) if has_pos else None
        lin_w, lin_b = (rest[0], rest[1]) if has_lin else (None, None)
        dev = row_ce.device
        lib = N.load()
        B = row_ce.shape[0]
        pue, _, DI, ld_ue = _f32_2d(ue, "user_embedding")
        g = g.contiguous().to(torch.float32)
        d_ce = torch.empty(B, dtype=torch.float32, device=dev)
        d_ue = torch.empty(B, DI, dtype=torch.float32, device=dev)
        d_pos = torch.empty_like(pos_table) if has_pos else None
        d_w, d_b = (torch.empty_like(lin_w), torch.empty_like(lin_b)) if has_lin else (None, None)
        N.check(lib.tt_debias_loss_bwd(ctx.mode, g.data_ptr(), row_ce.data_ptr(), B, position.data_ptr(), ctx.n_pos, pue,
                                       ld_ue, DI, N.ptr(lin_w), ws.data_ptr(), ws.numel(), d_ce.data_ptr(),
                                       d_ue.data_ptr(), DI, N.ptr(d_pos), N.ptr(d_w), N.ptr(d_b), N.stream()),
                "tt_debias_loss_bwd")
        return d_ce, None, None, None, d_ue, d_pos, d_w, d_b, None


# ----------------------------------------------------------------- history encoder
def fold_weights(w_in, b_in, w_po, b_po, slot: str = "ws"):
    """(W_in W_o, W_in b_o + b_in): the in-projection of a layer that reads the previous layer's attention CONTEXT --
    [3D, D] x [D, D], 12.6 MFLOP at D = 128, instead of a [B*H, D] x [D, D] out-projection (and its d_ctx / dW_out products
    in the backward).  (One hand-written launch per direction was tried in round 5 and was 4x slower than these two small
    library products: profiles/HISTORY.md.)"""
    D = w_po.shape[0]
    dev = w_in.device
    w_eff = torch.empty(3 * D, D, dtype=torch.float32, device=dev)
    gemm(N.TT_GEMM_NN, w_in, w_po, w_eff, 3 * D, D, D, slot=slot)
    b_eff = torch.empty(1, 3 * D, dtype=torch.float32, device=dev)
    gemm(N.TT_GEMM_NT, b_po.view(1, D), w_in, b_eff, 1, 3 * D, D, bias=b_in, slot=slot)
    return w_eff, b_eff.view(-1)


def fold_weight_grads(G, s, w_in, w_po, b_po, dW_in, dW_po, db_po) -> None:
    """Gradients of a composed boundary (G = dL/dW_eff, s = dL/db_eff) back to the two layers' own parameters:
    dW_in = G W_o^T + s (x) b_o,  dW_o = W_in^T G,  db_o = W_in^T s   (db_in = s)."""
    D = w_po.shape[0]
    gemm(N.TT_GEMM_NT, G, w_po, dW_in, 3 * D, D, D)
    gemm(N.TT_GEMM_NT, s.view(3 * D, 1), b_po.view(D, 1), dW_in, 3 * D, D, 1, accumulate=True)
    gemm(N.TT_GEMM_TN, w_in, G, dW_po, D, D, 3 * D)
    gemm(N.TT_GEMM_NN, s.view(1, 3 * D), w_in, db_po.view(1, D), 1, D, 3 * D)


def _attn_fwd(qkv, B, H, D, heads):
    ctx_t = torch.empty(B * H, D, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, heads, H, dtype=torch.float32, device=qkv.device)
    N.check(N.load().tt_attn_fwd(qkv.data_ptr(), B, H, D, heads, ctx_t.data_ptr(), lse.data_ptr(), N.stream()),
            "tt_attn_fwd")
    return ctx_t, lse


class HistoryEncoder(_LookupFunction):
    """UserHistoryEncoder.forward (ref:src/user_history_encoder.py:80-121), optionally
    fused with the history id lookup (ref:src/two_tower_with_user_history_encoder.py:105).

    source = embedding table [N, D] with ids [B, H]   (ids given), or
    source = embedded history [B, H, D]               (ids None).
    Returns [B, 2, D]: slot 0 = row 0 after L attention layers, slot 1 = mean of raw rows.
    The last layer's out-projection is evaluated for row 0 only (the only row consumed)."""

    @staticmethod
    def forward(ctx, source, ids, pe, heads: int, *layer_params):
        dev = N.require_device(source, ids, pe, *layer_params)
        L = len(layer_params) // 4
        lib = N.load()
        ctx.lookup_index = None
        if ids is not None:
            if ids.dtype not in (torch.int64, torch.int32):  # nn.Embedding accepts exactly these two
                raise TypeError(f"user_history ids must be int64 or int32, got {ids.dtype}")
            ids = ids.to(torch.int64).contiguous()  # tt_hist_embed_pool reads int64
            B, H = ids.shape
            D = source.shape[1]
            src, row_ids, ctx.lookup_index = lookup_source(source, ids, _recording(ctx))
            n_rows = src.shape[0]
            gather_ids = row_ids.reshape(B, H)
        else:
            src = source.contiguous()
            B, H, D = src.shape
            n_rows = 0
        out = torch.empty(B, 2, D, dtype=torch.float32, device=dev)
        x = torch.empty(B * H, D, dtype=torch.float32, device=dev)
        pooled = out[:, 1, :]
        N.check(lib.tt_hist_embed_pool(src.data_ptr(), n_rows, D, N.ptr(gather_ids if ids is not None else None), B, H,
                                       N.ptr(pe), x.data_ptr(),
                                       pooled.data_ptr(), 2 * D, N.oob.flag(dev).data_ptr(), N.stream()),
                "tt_hist_embed_pool")
        sweep_opt = None
        if ids is not None:
            # the step's big gather is queued: a table sweep the optimiser held back for it may start now (optim.py) -- unless
            # the sweep steps over this step's rows (marked schedule): nothing orders it against the lookups then, and it
            # starts behind the first in-projection below
            ref = getattr(source, "_tt_optimizer", None)
            opt = ref() if ref is not None else None
            if opt is not None:
                if opt.held_sweep_is_marked():
                    sweep_opt = opt
                else:
                    opt.release_sweep()
        saved: List[torch.Tensor] = []
        # the last layer is consumed at row 0 only: one query per (sample, head), K / V projections folded into two D-wide
        # vectors per (sample, head) (csrc/encoder_last.hip)
        collapsed_last = (L > 0 and not _ENC_GENERIC and H <= 64 and D // heads <= 64
                          and bool(lib.tt_enc_last_supported(H, D, heads)) and x.data_ptr() % 16 == 0)
        # ... run on the second-to-last layer's CONTEXT with composed weights (like every other layer boundary below)
        fold_last = collapsed_last and L >= 2
        dh = D // max(heads, 1)
        if L > 0 and (H > 64 or dh not in (16, 32, 64) or D % 4):
            note_generic("history-encoder attention",
                         f"the matrix-core kernels take H <= 64 and head width in {{16, 32, 64}}; got H = {H}, head width = {dh}: "
                         "VALU attention" + ("" if collapsed_last else ", last layer computed for every position"))
        fold = not _ENC_GENERIC  # out-projection of layer l composed with the in-projection of layer l + 1

        def folded_in(l):  # layer l (a full layer, not the first) reads the previous layer's context through composed weights
            return fold and 1 <= l < L and not (l == L - 1 and collapsed_last)

        folded_w = {}
        # The composed weights depend on nothing but the parameters: all boundaries' products (two 10-us library launches
        # each, 20 - 25 us each next to the table sweep) are queued on the third stream HERE, underneath the history gather
        # and the first layer, instead of between the layers on the main stream; a layer waits for its own pair's event.
        # (Allocated on the third stream, read on the main one until the backward pass ends: every later use of the third
        # stream starts by waiting for the main one, see AuxFork.)
        composed = {}
        will_fold = [l for l in range(L) if folded_in(l) or (l == L - 1 and fold_last)]
        if will_fold and _FOLD_ASIDE and dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            top, aux_s = torch.cuda.Event(), N.aux_stream(dev)
            top.record()
            aux_s.wait_event(top)
            with torch.cuda.stream(aux_s):
                for l in will_fold:
                    w_po, b_po = layer_params[4 * (l - 1) + 2], layer_params[4 * (l - 1) + 3]
                    we, be = fold_weights(layer_params[4 * l], layer_params[4 * l + 1], w_po, b_po, slot="ws_side_f")
                    ready = torch.cuda.Event()
                    ready.record(aux_s)
                    composed[l] = (we, be, ready)

        def composed_weights(l, w_in, b_in):
            if l in composed:
                we, be, ready = composed.pop(l)
                torch.cuda.current_stream(dev).wait_event(ready)
                return we, be
            return fold_weights(w_in, b_in, layer_params[4 * (l - 1) + 2], layer_params[4 * (l - 1) + 3])

        for l in range(L):
            w_in, b_in, w_out, b_out = layer_params[4 * l: 4 * l + 4]
            if l == L - 1 and collapsed_last:
                # the last layer is consumed at row 0 only and its K / V projections fold into two D-wide vectors per
                # (sample, head): x is read once, no [B*H, 2D] projection (csrc/encoder_last.hip)
                q0 = torch.empty(B, D, dtype=torch.float32, device=dev)
                tq = torch.empty(B, heads, D, dtype=torch.float32, device=dev)
                probs = torch.empty(B, heads, H, dtype=torch.float32, device=dev)
                xbar = torch.empty(B, heads, D, dtype=torch.float32, device=dev)
                ctx0 = torch.empty(B, D, dtype=torch.float32, device=dev)
                w_in_c, w_out_c, b_in_c = w_in.contiguous(), w_out.contiguous(), b_in.contiguous()
                if fold_last:
                    # x is the previous layer's CONTEXT c.  q, k, v of this layer are linear in x = c W_o^T + b_o, so the layer
                    # runs on c with W_eff = W_in W_o, b_eff = W_in b_o + b_in -- the same kernels as for a first layer
                    w_in_c, b_in_c = composed_weights(l, w_in, b_in)
                    folded_w[l] = w_in_c
                N.check(lib.tt_enc_last_fwd(x.data_ptr(), B, H, D, heads, w_in_c.data_ptr(), b_in_c.data_ptr(),
                                            w_out_c.data_ptr(), b_out.contiguous().data_ptr(), out.data_ptr(), 2 * D,
                                            q0.data_ptr(), tq.data_ptr(), probs.data_ptr(), xbar.data_ptr(), ctx0.data_ptr(),
                                            N.stream()), "tt_enc_last_fwd")
                saved += [x, q0, tq, probs, xbar, ctx0]
                continue
            qkv = torch.empty(B * H, 3 * D, dtype=torch.float32, device=dev)
            if folded_in(l):
                # x is the previous layer's CONTEXT c: x_l = c W_o^T + b_o never exists, the two Linear maps are composed --
                #   qkv = c (W_in W_o)^T + (W_in b_o + b_in)          ([3D, D] x [D, D]: 12.6 MFLOP instead of a [B*H, D] x [D, D]
                # out-projection, and its d_ctx / dW_out products in the backward)
                w_eff, b_eff = composed_weights(l, w_in, b_in)
                gemm(N.TT_GEMM_NT, x, w_eff, qkv, B * H, 3 * D, D, bias=b_eff)
                folded_w[l] = w_eff
            else:
                gemm(N.TT_GEMM_NT, x, w_in, qkv, B * H, 3 * D, D, bias=b_in)
            if sweep_opt is not None:
                # Measured over every launch of the C3 step as the release point, each at its best sweep width
                # (tools/sweep_release_scan.py, profiles/r06_sweep_release_scan.txt): behind the history gather 3.14-3.15 ms,
                # behind THIS product 3.11-3.12, behind the attention kernel 3.22-3.25, behind the second layer's product
                # 3.18-3.22, later 3.2-3.5; parked rows (which start the sweep about here: the moments' gather ran in front of
                # it) 3.32-3.37
                started = torch.cuda.Event()
                started.record()
                sweep_opt.release_sweep(after=started)
                sweep_opt = None
            ctx_t, lse = _attn_fwd(qkv, B, H, D, heads)
            saved += [x, qkv, ctx_t, lse]
            if (l + 2 == L and fold_last) or folded_in(l + 1):
                x = ctx_t  # no out-projection here: the next layer takes the context (fold above / csrc/encoder_last.hip, PREV)
            elif l + 1 < L:
                x = torch.empty(B * H, D, dtype=torch.float32, device=dev)
                gemm(N.TT_GEMM_NT, ctx_t, w_out, x, B * H, D, D, bias=b_out)
            else:  # rows b*H + 0 only, straight into out[:, 0, :]
                gemm(N.TT_GEMM_NT, ctx_t.view(B, H * D)[:, :D], w_out, out[:, 0, :], B, D, D, bias=b_out)
        if L == 0:
            out[:, 0, :].copy_(x.view(B, H, D)[:, 0, :])
        if sweep_opt is not None:  # no full layer ran
            sweep_opt.release_sweep()
        ctx.dims = (B, H, D, L, heads)
        ctx.collapsed_last = collapsed_last
        ctx.folded = {l: w for l, w in folded_w.items()}  # layer -> its composed in-projection weight W_in W_o(prev)
        ctx.layer_leaves = tuple(layer_params)  # the Parameter objects themselves (see ops.run_on_side: `leaves`)
        if L > 0 and _caller_grad_mode[0] and any(ctx.needs_input_grad[4:]):  # (grad mode is always off in here: _recording)
            _side_state["encoder"] = True
        ctx.table = source if ids is not None else None
        ctx.has_ids = ids is not None
        ctx.save_for_backward(ids, *layer_params, *saved)
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, H, D, L, heads = ctx.dims
        t = ctx.saved_tensors
        ids = t[0]
        layer_params = t[1: 1 + 4 * L]
        saved = t[1 + 4 * L:]
        dev = d_out.device
        lib = N.load()
        d_out = d_out.contiguous()
        d_recent, d_pooled = d_out[:, 0, :], d_out[:, 1, :]
        grads: List[Optional[torch.Tensor]] = [None] * (4 * L)
        dx = None  # gradient wrt the current layer's OUTPUT x_{l+1}, [B*H, D]
        prev_out_grads = None
        # (fn, event of the side-stream product it consumes or None): the composed boundaries' small weight-gradient products.
        # They run at the END of this backward, IN LINE on the main stream: by then the main stream has nothing left to do
        # but wait for the side stream's streaming weight gradients, and the side stream has those still to run
        tail_jobs = []
        pool_done = False
        leaf_params = ctx.layer_leaves

        def wgrad(dy, xin, dW, tag, l):  # off the critical path: see run_on_side
            db = torch.empty(dW.shape[0], dtype=torch.float32, device=dev)
            k = 4 * l + (0 if tag == "i" else 2)  # the (weight, bias) pair this product is the gradient of
            run_on_side(dev, lambda on_side: gemm_tn_colsum(dy, xin, dW, db=db, slot=("ws_side_" + tag) if on_side else "ws"),
                        hold=(dy, xin), leaves=leaf_params[k: k + 2])
            return db

        for l in reversed(range(L)):
            w_in, b_in, w_out, b_out = layer_params[4 * l: 4 * l + 4]
            if l == L - 1 and ctx.collapsed_last:
                x, q0, tq, probs, xbar, ctx0 = saved[4 * l: 4 * l + 6]
                dx = torch.empty(B * H, D, dtype=torch.float32, device=dev)
                dW_in = torch.empty(3 * D, D, dtype=torch.float32, device=dev)
                db_in = torch.empty(3 * D, dtype=torch.float32, device=dev)
                dW_out = torch.empty(D, D, dtype=torch.float32, device=dev)
                db_out = torch.empty(D, dtype=torch.float32, device=dev)
                # one workspace for both halves, whichever stream the weight half runs on (the next user of the slot is the
                # next step's backward pass, behind the end-of-backward join)
                wsp, wsn = _ws(dev, lib.tt_enc_last_bwd_workspace_bytes(B, H, D, heads), "enc_last")
                w_used = ctx.folded.get(l, w_in)  # the composed W_in W_o(prev) when the forward ran on the previous layer's context
                N.check(lib.tt_enc_last_bwd_data(x.data_ptr(), B, H, D, heads, w_used.contiguous().data_ptr(),
                                                 w_out.contiguous().data_ptr(), d_recent.data_ptr(), 2 * D, tq.data_ptr(),
                                                 probs.data_ptr(), dx.data_ptr(), wsp, wsn, N.stream()), "tt_enc_last_bwd_data")

                # the weight half ([D, 32] x [32, D] partial products per 32 samples + their reduction) feeds nothing but the
                # optimiser: third stream (it was the tail of the data kernels: 0.1 ms of the step's critical path)
                def last_weights(on_side, x=x, q0=q0, xbar=xbar, ctx0=ctx0, dW_in=dW_in, db_in=db_in, dW_out=dW_out, db_out=db_out,
                                 wsp=wsp, wsn=wsn):
                    N.check(lib.tt_enc_last_bwd_weights(x.data_ptr(), B, H, D, heads, d_recent.data_ptr(), 2 * D, q0.data_ptr(),
                                                        xbar.data_ptr(), ctx0.data_ptr(), dW_in.data_ptr(), db_in.data_ptr(),
                                                        dW_out.data_ptr(), db_out.data_ptr(), wsp, wsn, N.stream()),
                            "tt_enc_last_bwd_weights")

                # leaves: the Parameters whose gradients the side job writes -- with a composed W_in, dW_in here is the private
                # G (turned into the two layers' gradients by a tail job on the main stream), so w_in is not among them
                lw = list(leaf_params[4 * l + 1: 4 * l + 4]) + ([] if l in ctx.folded else [leaf_params[4 * l]])
                w_aside = False
                if _EL_WGRAD_SIDE:
                    w_aside = run_on_side(dev, last_weights, hold=(x, q0, xbar, ctx0, d_out) + ((dW_in,) if l in ctx.folded else ()),
                                          leaves=lw)
                else:
                    last_weights(False)
                w_done = None
                if w_aside and l in ctx.folded:  # the tail job below reads G on the main stream
                    w_done = torch.cuda.Event()
                    w_done.record(N.aux_stream(dev))
                if l in ctx.folded:
                    # dW_in / db_in above are the gradients of (W_eff, b_eff): G and s.  Back to the two layers' own parameters
                    # (the four small products of a composed boundary, see the full layers below):
                    #   dW_in = G W_o^T + s (x) b_o     db_in = s     dW_o = W_in^T G     db_o = W_in^T s
                    w_po, b_po = layer_params[4 * (l - 1) + 2], layer_params[4 * (l - 1) + 3]
                    G, dW_in = dW_in, torch.empty(3 * D, D, dtype=torch.float32, device=dev)
                    dW_po = torch.empty(D, D, dtype=torch.float32, device=dev)
                    db_po = torch.empty(D, dtype=torch.float32, device=dev)

                    def last_folded_weights(on_side, G=G, db_in=db_in, dW_in=dW_in, dW_po=dW_po, db_po=db_po, w_po=w_po, b_po=b_po,
                                            w_in=w_in):
                        fold_weight_grads(G, db_in, w_in, w_po, b_po, dW_in, dW_po, db_po)

                    # (issued here on the side stream, these four 10-us products sat behind the persistent attention-backward
                    # workgroups for 0.45 ms and pushed the streaming weight gradients that much later)
                    tail_jobs.append((last_folded_weights, w_done))
                    prev_out_grads = (dW_po, db_po)  # dx is the gradient of the previous layer's CONTEXT
                grads[4 * l: 4 * l + 4] = [dW_in, db_in, dW_out, db_out]
                continue
            x, qkv, ctx_t, lse = saved[4 * l: 4 * l + 4]
            dW_out = torch.empty(D, D, dtype=torch.float32, device=dev)
            if prev_out_grads is not None:  # the layer after this one took the context: its backward produced these
                dW_out, db_out = prev_out_grads
                d_ctx = dx
                prev_out_grads = None
            elif l == L - 1:
                rows0 = ctx_t.view(B, H * D)[:, :D]
                _, db_out = gemm_tn_colsum(d_recent, rows0, dW_out)
                d_ctx = torch.zeros(B * H, D, dtype=torch.float32, device=dev)
                gemm(N.TT_GEMM_NN, d_recent, w_out, d_ctx.view(B, H * D)[:, :D], B, D, D)
            else:
                db_out = wgrad(dx, ctx_t, dW_out, "o", l)
                d_ctx = torch.empty(B * H, D, dtype=torch.float32, device=dev)
                gemm(N.TT_GEMM_NN, dx, w_out, d_ctx, B * H, D, D)
            d_qkv = torch.empty(B * H, 3 * D, dtype=torch.float32, device=dev)
            N.check(lib.tt_attn_bwd(qkv.data_ptr(), ctx_t.data_ptr(), lse.data_ptr(), d_ctx.data_ptr(), B, H, D,
                                    heads, d_qkv.data_ptr(), N.stream()), "tt_attn_bwd")
            dW_in = torch.empty(3 * D, D, dtype=torch.float32, device=dev)
            if l in ctx.folded:
                # x is the previous layer's context c and qkv = c W_eff^T + b_eff with W_eff = W_in W_o, b_eff = W_in b_o + b_in:
                #   G = dQKV^T c, s = colsum(dQKV)      dW_in = G W_o^T + s (x) b_o      db_in = s
                #   dW_o = W_in^T G                      db_o = W_in^T s                  d_c = dQKV W_eff  (the data path)
                w_po, b_po = layer_params[4 * (l - 1) + 2], layer_params[4 * (l - 1) + 3]
                w_eff = ctx.folded[l]
                G = torch.empty(3 * D, D, dtype=torch.float32, device=dev)
                db_in = torch.empty(3 * D, dtype=torch.float32, device=dev)
                dW_po = torch.empty(D, D, dtype=torch.float32, device=dev)
                db_po = torch.empty(D, dtype=torch.float32, device=dev)

                def folded_G(on_side, d_qkv=d_qkv, x=x, G=G, db_in=db_in):
                    gemm_tn_colsum(d_qkv, x, G, db=db_in, slot="ws_side_i" if on_side else "ws")

                def folded_weights(on_side, G=G, db_in=db_in, dW_in=dW_in, dW_po=dW_po, db_po=db_po, w_po=w_po, b_po=b_po, w_in=w_in):
                    fold_weight_grads(G, db_in, w_in, w_po, b_po, dW_in, dW_po, db_po)

                # the streaming product G is queued BEFORE the data-path product below (the side stream starts where the main
                # one stands now); the four small products that turn G into weight gradients wait until the END of this
                # backward -- in line they sat on the side stream between this layer's G and the next one's, next to the
                # data path's heaviest kernels, 245 us for 60 us of work, and the side stream finished 250 us after the main one
                # (held: INPUTS and the private G only -- never db_in, which is returned as in_proj_bias's gradient: a held
                # output is CLONED by AccumulateGrad on the main stream before the side stream has written it)
                hold_f = (d_qkv, x, G, w_po, b_po, w_in, w_eff)
                leaves_f = list(leaf_params[4 * l: 4 * l + 2]) + list(leaf_params[4 * (l - 1) + 2: 4 * (l - 1) + 4])
                aside = run_on_side(dev, folded_G, hold=hold_f, leaves=leaves_f)
                g_done = None
                if aside:  # G is written on the side stream: the tail job (main stream) waits for exactly that
                    g_done = torch.cuda.Event()
                    g_done.record(N.aux_stream(dev))
                tail_jobs.append((folded_weights, g_done))
                dx = torch.empty(B * H, D, dtype=torch.float32, device=dev)
                gemm(N.TT_GEMM_NN, d_qkv, w_eff, dx, B * H, D, 3 * D)
                prev_out_grads = (dW_po, db_po)
            else:
                db_in = wgrad(d_qkv, x, dW_in, "i", l)
                dx = torch.empty(B * H, D, dtype=torch.float32, device=dev)
                rc = N.TT_E_UNSUPPORTED
                if l == 0 and not _ENC_GENERIC and w_in.is_contiguous():
                    rc = lib.tt_hist_dx_pool_bwd(d_qkv.data_ptr(), w_in.data_ptr(), B, H, D, d_pooled.data_ptr(), 2 * D, dx.data_ptr(),
                                                 N.stream())
                if rc == 0:
                    pool_done = True  # the mean pool's backward rode in this product's epilogue (csrc/gemm_ws16.hip, POOL)
                elif rc == N.TT_E_UNSUPPORTED:  # shape not taken: the product, then a read-modify-write pass over dx
                    gemm(N.TT_GEMM_NN, d_qkv, w_in, dx, B * H, D, 3 * D)
                else:
                    N.check(rc, "tt_hist_dx_pool_bwd")
            grads[4 * l: 4 * l + 4] = [dW_in, db_in, dW_out, db_out]
        for fn, g_done in tail_jobs:
            if g_done is not None:
                torch.cuda.current_stream().wait_event(g_done)
            fn(False)
        if dx is None:  # L == 0: slot 0 is row 0 of (x + pe)
            dx = torch.zeros(B * H, D, dtype=torch.float32, device=dev)
            dx.view(B, H, D)[:, 0, :].copy_(d_recent)
        if not pool_done:
            N.check(lib.tt_hist_pool_bwd(dx.data_ptr(), B, H, D, d_pooled.data_ptr(), 2 * D, N.stream()),
                    "tt_hist_pool_bwd")
        d_source = None
        if ctx.needs_input_grad[0]:
            if ctx.has_ids:
                d_source = _route_table_grad(ctx.table, ids.reshape(-1), dx, ctx.lookup_index)
            else:
                d_source = dx.view(B, H, D)
        return (d_source, None, None, None, *grads)


# ----------------------------------------------------------------- MIPS
def mips_split_rows(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 [R, 128] -> ([R, 256] fp16 = the two-term split [h | l] of x * scale, scale [1]): the operand form of the
    EXPLORATORY TT_F16X2 scoring (tt_mips_split_rows; csrc/mips.hip, csrc/ce_f16x2.hip)."""
    dev = N.require_device(x)
    lib = N.load()
    x = x.detach().contiguous()
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != 128:
        raise TypeError("split-fp16 scoring takes float32 [rows, 128]")
    out = torch.empty(x.shape[0], 2 * x.shape[1], dtype=torch.float16, device=dev)
    scale = torch.empty(1, dtype=torch.float32, device=dev)
    wsp, wsn = _ws(dev, 256, "mips_split")
    N.check(lib.tt_mips_split_rows(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), scale.data_ptr(), wsp, wsn, N.stream()),
            "tt_mips_split_rows")
    return out, scale


def mips_topk(query: torch.Tensor, corpus: torch.Tensor, k: int,
              split16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch.topk(query @ corpus.T, k) with (score desc, index asc) order
    (ref:src/baseline_mips_module.py:57-61).  corpus fp32 or bf16 [C, D]; query fp32 [B, D].
    split16 = mips_split_rows(corpus) (EXPLORATORY, fp32 corpus with D = 128): score on the fp16 matrix pipe from two-term
    splits of both operands -- fp32-grade scores, the fp32 path's contract, about 2.5x its speed."""
    dev = N.require_device(query, corpus)
    lib = N.load()
    if split16 is not None and corpus.dtype == torch.float32 and corpus.shape[1] == 128 and query.dtype == torch.float32 \
            and query.dim() == 2 and query.shape[1] == 128 and 0 < k <= corpus.shape[0]:
        c16, c_scale = split16
        if c16.shape != (corpus.shape[0], 256) or c16.dtype != torch.float16:
            raise ValueError("split16 does not belong to this corpus")
        q16, q_scale = mips_split_rows(query)
        B, Cn = query.shape[0], corpus.shape[0]
        idx = torch.empty(B, k, dtype=torch.int64, device=dev)
        scores = torch.empty(B, k, dtype=torch.float32, device=dev)
        wsp, wsn = _ws(dev, lib.tt_mips_workspace_bytes(B, Cn, 128, k, N.TT_F16X2), "mips")
        N.check(lib.tt_mips_topk(q16.data_ptr(), c16.data_ptr(), N.TT_F16X2, B, Cn, 128, k, idx.data_ptr(), scores.data_ptr(),
                                 wsp, wsn, N.stream()), "tt_mips_topk")
        N.check(lib.tt_mips_unscale(scores.data_ptr(), B * k, q_scale.data_ptr(), c_scale.data_ptr(), N.stream()), "tt_mips_unscale")
        return idx, scores
    query = query.detach()
    if query.dtype != torch.float32 or query.dim() != 2:
        raise TypeError("query_embedding must be a 2-D float32 tensor")
    B, D = query.shape
    Cn = corpus.shape[0]
    if corpus.shape[1] != D:
        raise RuntimeError(f"query dim {D} != corpus dim {corpus.shape[1]}")  # torch.matmul would raise too
    if not (0 < k <= Cn):
        raise RuntimeError("selected index k out of range")  # torch.topk's message
    corpus = corpus.contiguous()
    if D > 128:
        note_generic("MIPS top-K", f"D = {D} > 128: score slabs from the library GEMM instead of the register-stationary pass")
    if corpus.dtype == torch.bfloat16:
        dtype = N.TT_BF16
        q = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
        N.check(lib.tt_f32_to_bf16(query.contiguous().data_ptr(), q.data_ptr(), B * D, N.stream()), "tt_f32_to_bf16")
    elif corpus.dtype == torch.float32:
        dtype = N.TT_F32
        q = query.contiguous()
    else:
        raise TypeError("corpus must be float32 or bfloat16")
    idx = torch.empty(B, k, dtype=torch.int64, device=dev)
    scores = torch.empty(B, k, dtype=torch.float32, device=dev)
    wsp, wsn = _ws(dev, lib.tt_mips_workspace_bytes(B, Cn, D, k, dtype), "mips")
    N.check(lib.tt_mips_topk(q.data_ptr(), corpus.data_ptr(), dtype, B, Cn, D, k, idx.data_ptr(),
                             scores.data_ptr(), wsp, wsn, N.stream()), "tt_mips_topk")
    return idx, scores


def mips_merge(scores: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per query, the k best of the [B, n_cand] (score, global index) candidates."""
    dev = N.require_device(scores, idx)
    lib = N.load()
    scores, idx = scores.contiguous(), idx.contiguous()
    B, n_cand = scores.shape
    out_idx = torch.empty(B, k, dtype=torch.int64, device=dev)
    out_sc = torch.empty(B, k, dtype=torch.float32, device=dev)
    wsp, wsn = _ws(dev, lib.tt_mips_merge_workspace_bytes(B, n_cand), "mips")
    N.check(lib.tt_mips_merge(scores.data_ptr(), idx.data_ptr(), B, n_cand, k, out_idx.data_ptr(),
                              out_sc.data_ptr(), wsp, wsn, N.stream()), "tt_mips_merge")
    return out_idx, out_sc


def gather_corpus_rows(corpus: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """corpus[idx] -> [B, K, D] fp32 (ref:src/baseline_mips_module.py:63-69)."""
    dev = N.require_device(corpus, idx)
    B, K = idx.shape
    D = corpus.shape[1]
    out = torch.empty(B * K, D, dtype=torch.float32, device=dev)
    flat = idx.reshape(-1).contiguous()
    if corpus.dtype == torch.bfloat16:
        N.check(N.load().tt_gather_rows_bf16(corpus.data_ptr(), corpus.shape[0], D, flat.data_ptr(), B * K,
                                             out.data_ptr(), D, N.oob.flag(dev).data_ptr(), N.stream()),
                "tt_gather_rows_bf16")
    else:
        gather_rows_into(corpus, flat, out)
    return out.view(B, K, D)
