"""Row-sharded multi-GPU train step (one process per GPU, RCCL over xGMI through
torch.distributed; backend "nccl" is RCCL on ROCm).

The reference has no parallelism of any kind; this is new design,
constrained only by parity: W ranks with B rows each compute exactly the reference's loss
and update on the CONCATENATED batch of W*B rows ("global in-batch negatives").

Partitioning
  * both embedding tables are split into W contiguous row blocks; a rank owns the block and
    its Adam moments, and sweeps only its block (the HBM-bound part scales 1/W, no comms);
  * the batch is split by rank; dense MLP / tower parameters are replicated.
Exchanges per step, every size known to the host before the step starts:
  1. lookups -- padded fixed-capacity all-to-all (`routing="alltoall"`, default).  A rank's ids are
     bucketed by owner in one stable counting pass (csrc/route.hip); each owner is sent only
     ITS ids, `cap` slots per peer, and returns the rows in the same slots:
        ids   all_to_all  [W, cap] int64          rows  all_to_all  [W, cap, D]
     `cap` = the largest (requester, owner) bucket over all ranks, rounded up to 64 -- exact, so
     no id is ever dropped and there is no overflow path.  It is all-reduced (MAX, 3 ints) ONE STEP
     AHEAD from the next batch's ids (`step(batch, next_batch)`), so the host never waits for it;
     without `next_batch` the step synchronises once on that scalar.
  2. item embeddings                         all_gather      [B, D] -> [W*B, D]
  3. max of the value weights, loss          all_reduce      scalars
  4. partial dI over the gathered items      reduce_scatter  [W*B, D] -> [B, D]
  5. dense-parameter gradients (flat buffer) all_reduce      ~0.5 MB
  6. embedding-row gradients: back through the lookup's slots   all_to_all  [W, cap, D]
     (each owner receives exactly the gradient rows of the ids it served, in the order it
     served them; padding slots carry the sentinel row the Adam kernels skip).
Per rank and step at W = 8, B = 8192, D = 128 the lookup traffic is 4 x W*cap*D*4 B ~ 4 x 4.7 MB
(was 4 x 33.5 MB with the all-gather scheme below); with the history model B*H rows = 105 MB per
direction (was 839 MB).  `trainer.comm_bytes` logs the bytes each exchange sends per step.
`routing="allgather"` (TT_ROUTE=allgather) keeps round 1's fixed-size scheme as the A/B partner:
all_gather of ids, every owner gathers W*B rows (zeros for foreign ids), reduce_scatter; row
gradients all_gather'ed to every owner.
Every rank then runs the dense-exact Adam of optim.py on its block, with the zero-gradient
sweep on a side stream underneath steps 1-6.

The arithmetic is delegated to a `backend` object.  The product backend is `HipBackend`
(libtt_hotpath.so); it is the default and the only one shipped.  tests/ inject a CPU
restatement to exercise the routing / collective logic under gloo without a GPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

TOWER_KEYS = ("features_arch.0.weight", "features_arch.0.bias", "features_arch.2.weight",
              "features_arch.2.bias", "tower_arch.weight", "tower_arch.bias")


# ----------------------------------------------------------------- collectives
# Two transports behind the same helpers: torch.distributed's process group (default; "nccl" = RCCL), or the
# C ABI's own tt_comm_* (comm.NativeComm, `use_native_transport`) -- RCCL bound by libtt_hotpath.so itself.
_NATIVE = None  # comm.NativeComm
_COMM_STREAM: Optional[torch.cuda.Stream] = None
_NATIVE_DTYPES = (torch.float32, torch.int32, torch.int64, torch.uint8)


def use_native_transport(comm) -> None:
    """Route every device-side collective of this module through `comm` (a comm.NativeComm), or back through
    torch.distributed with None.  The process group stays what creates / synchronises the ranks."""
    global _NATIVE, _COMM_STREAM
    if _NATIVE is not None and _NATIVE is not comm:
        _NATIVE.close()
    _NATIVE = comm
    _COMM_STREAM = torch.cuda.Stream(device=comm.device) if comm is not None else None
    if comm is not None and not getattr(use_native_transport, "_atexit", False):
        import atexit
        atexit.register(lambda: use_native_transport(None))  # the communicator goes before the process group
        use_native_transport._atexit = True


# TT_COMM_FORCE_ASYNC=1 (tests, 1-GPU boxes): take the RCCL code paths -- async collectives on the process group's
# stream / tt_comm_* on the communication stream -- at world size 1 as well, where every collective is the identity.
# The paths a multi-GPU node runs are then executed, with the real message sizes, on the box that has one device.
def _force_async() -> bool:
    return os.environ.get("TT_COMM_FORCE_ASYNC") is not None


def _native(x: torch.Tensor) -> bool:
    return (_NATIVE is not None and x.is_cuda and x.dtype in _NATIVE_DTYPES
            and (dist.get_world_size() > 1 or _force_async()))


def _native_op(op) -> int:
    from . import _native as N
    if op == dist.ReduceOp.SUM:
        return N.TT_COMM_SUM
    if op == dist.ReduceOp.MAX:
        return N.TT_COMM_MAX
    raise ValueError("native transport: SUM and MAX reductions only")


def _is_gloo() -> bool:
    return dist.get_backend() == "gloo"


def _host_staged(x: torch.Tensor) -> bool:
    """gloo moves host memory only.  Device tensors under gloo (several ranks sharing ONE GPU, used
    by tests/test_gpu_sharded.py to run the HIP backend at world size > 1 on a 1-GPU box) are
    staged through the host; under RCCL nothing is staged."""
    return _is_gloo() and x.device.type != "cpu"


def all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    if _native(x):
        return _NATIVE.all_gather(x)
    if _host_staged(x):
        return all_gather_rows(x.cpu()).to(x.device)
    out = x.new_empty((dist.get_world_size() * x.shape[0],) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, x.contiguous())
    return out


def reduce_scatter_rows(x: torch.Tensor) -> torch.Tensor:
    if _native(x):
        return _NATIVE.reduce_scatter(x)
    if _host_staged(x):
        return reduce_scatter_rows(x.cpu()).to(x.device)
    W = dist.get_world_size()
    out = x.new_empty((x.shape[0] // W,) + tuple(x.shape[1:]))
    if _is_gloo():  # gloo has no reduce_scatter: all_reduce + slice (tests only)
        y = x.clone()
        dist.all_reduce(y)
        r = dist.get_rank()
        out.copy_(y[r * out.shape[0]:(r + 1) * out.shape[0]])
    else:
        dist.reduce_scatter_tensor(out, x.contiguous())
    return out


# A/B (opt-in): user tower backward on the third stream underneath the dI logits kernel.  Measured round 4 on the emulated
# W = 8 step: the towers' 0.1 ms leave the tail, the logits kernel they now share the chip with takes as much longer --
# 4.237 vs 4.246 ms per step; not the default.
_UTOWER_EARLY = os.environ.get("TT_SHARDED_EARLY_UTOWER") is not None
_LOSS_KERNELS = os.environ.get("TT_SHARDED_TORCH_LOSS") is None  # A/B: the value-weight tail as two kernels
_WGRAD_ASIDE = os.environ.get("TT_SHARDED_WGRAD_MAIN") is None  # A/B: tower weight gradients on the third stream
_CE_F16X2 = os.environ.get("TT_CE_F16X2") is not None  # exploratory: split-fp16 logits kernels (HipBackend.ce_fwd)
_CE16_KEEP = os.environ.get("TT_CE16_KEEP") is not None  # A/B: that pair with kept logits (its first form) instead of recomputed ones
# Opt-in (TT_SHARDED_PLAN_ASIDE=1): the NEXT batch's route plan (owner histogram + scan per lookup, the MAX all-reduce
# of the bucket sizes, their copy to the host) on the library's third stream at the very top of the step instead of on
# the main stream after the lookups -- eight small launches leave the critical path: emulated W = 8 step 4.09 -> 4.05 ms
# (three A/B pairs on one box).  It has to be the EXISTING third stream (ops.run_on_side: a new HIP stream may share the
# sweep's hardware queue, which is what made the first attempt slower).  Not the default: it issues that all-reduce
# from a second stream while the lookups' exchanges are in flight on the first, a pattern no multi-GPU run has
# exercised yet, and 1 % is not worth a surprise there.
_PLAN_ASIDE = os.environ.get("TT_SHARDED_PLAN_ASIDE") is not None


# Per-exchange timing (bench.py's multi-rank line: `comm_ms`): None = off.  When a list, every exchange appends
# (tag, issued, wait_begin, wait_end[, comm_begin, comm_end]) CUDA events; `comm_timing_summary` turns them into, per
# tag and step:  span = issue -> result usable (what the exchange costs if NOTHING overlaps it), exposed = the time
# the compute stream actually stood still at `.wait()` (0 when the exchange finished underneath the kernels queued in
# between), and -- native transport, whose stream we own -- the collective's own duration on the wire.
_TIMING: Optional[list] = None


def comm_timing(on: bool) -> None:
    global _TIMING
    _TIMING = [] if on else None


def comm_timing_summary(steps: int) -> Dict[str, Dict[str, float]]:
    """Synchronises, then per tag: calls per step, span / exposed (/ wire) milliseconds per step."""
    out: Dict[str, Dict[str, float]] = {}
    if not _TIMING:
        return out
    torch.cuda.synchronize()
    for rec in _TIMING:
        tag, e_issue, w0, w1 = rec[:4]
        d = out.setdefault(tag, {"calls": 0, "span_ms": 0.0, "exposed_ms": 0.0})
        d["calls"] += 1
        d["span_ms"] += e_issue.elapsed_time(w1)
        d["exposed_ms"] += w0.elapsed_time(w1)
        if len(rec) > 4:
            d["wire_ms"] = d.get("wire_ms", 0.0) + rec[4].elapsed_time(rec[5])
    for d in out.values():
        for k in list(d):
            d[k] = round(d[k] / max(steps, 1), 4)
    _TIMING.clear()
    return out


def _tev() -> torch.cuda.Event:
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


class _Pending:
    """Result of a collective started with `*_start`: `.wait()` makes the CURRENT stream wait for it
    (RCCL: the collective runs on the process group's own stream -- or, native transport, on this module's
    communication stream -- meanwhile, so kernels launched in between overlap it) and returns the output."""

    __slots__ = ("out", "work", "keep", "tag", "issued", "wire")

    def __init__(self, out, work=None, keep=None, tag=None, issued=None, wire=None):
        self.out, self.work, self.keep = out, work, keep  # `keep`: the send buffer, alive until waited for
        self.tag, self.issued, self.wire = tag, issued, wire

    def wait(self) -> torch.Tensor:
        timing = _TIMING is not None and self.issued is not None
        w0 = _tev() if timing else None
        if self.work is not None:
            if isinstance(self.work, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(self.work)
            else:
                self.work.wait()
            self.work, self.keep = None, None
        if timing:
            _TIMING.append((self.tag, self.issued, w0, _tev()) + (tuple(self.wire) if self.wire else ()))
            self.issued = None
        return self.out


def _timed_sync(tag: str, fn, *args, **kw):
    """A blocking-style collective (the caller uses the result at once) under the same bookkeeping."""
    if _TIMING is None:
        return fn(*args, **kw)
    e0 = _tev()
    out = fn(*args, **kw)
    _TIMING.append((tag, e0, e0, _tev()))
    return out


def _native_start(fn, x: torch.Tensor, *args, tag=None) -> _Pending:
    """Run `fn(x, *args, stream=<communication stream>)` after everything queued so far on the current stream."""
    x = x.contiguous()
    timing = _TIMING is not None
    ready = torch.cuda.Event(enable_timing=timing)
    ready.record()
    _COMM_STREAM.wait_event(ready)
    wire = None
    if timing:
        c0 = torch.cuda.Event(enable_timing=True)
        c0.record(_COMM_STREAM)
    out = fn(x, *args, stream=_COMM_STREAM)
    done = torch.cuda.Event(enable_timing=timing)
    done.record(_COMM_STREAM)
    if timing:
        wire = (c0, done)
    return _Pending(out, done, x, tag, ready if timing else None, wire)


def _rccl_async(x: torch.Tensor) -> bool:
    return (dist.get_world_size() > 1 or _force_async()) and not _is_gloo() and x.is_cuda


def _issued():
    return _tev() if _TIMING is not None else None


def all_gather_rows_start(x: torch.Tensor, tag: str = "all_gather") -> _Pending:
    if _native(x):
        return _native_start(_NATIVE.all_gather, x, None, tag=tag)
    if not _rccl_async(x):  # gloo (tests) and world size 1: nothing to overlap with
        e = _issued()
        return _Pending(all_gather_rows(x) if dist.get_world_size() > 1 else x, tag=tag, issued=e)
    x = x.contiguous()
    out = x.new_empty((dist.get_world_size() * x.shape[0],) + tuple(x.shape[1:]))
    e = _issued()
    return _Pending(out, dist.all_gather_into_tensor(out, x, async_op=True), x, tag, e)


def reduce_scatter_rows_start(x: torch.Tensor, tag: str = "reduce_scatter") -> _Pending:
    if _native(x):
        from . import _native as N
        return _native_start(_NATIVE.reduce_scatter, x, None, N.TT_COMM_SUM, tag=tag)
    if not _rccl_async(x):
        e = _issued()
        return _Pending(reduce_scatter_rows(x) if dist.get_world_size() > 1 else x, tag=tag, issued=e)
    x = x.contiguous()
    out = x.new_empty((x.shape[0] // dist.get_world_size(),) + tuple(x.shape[1:]))
    e = _issued()
    return _Pending(out, dist.reduce_scatter_tensor(out, x, async_op=True), x, tag, e)


def all_reduce_start_(x: torch.Tensor, op=dist.ReduceOp.SUM, tag: str = "all_reduce") -> _Pending:
    if _native(x):
        return _native_start(_NATIVE.all_reduce_, x, _native_op(op), tag=tag)
    if not _rccl_async(x):
        e = _issued()
        return _Pending(all_reduce_(x, op=op) if dist.get_world_size() > 1 else x, tag=tag, issued=e)
    e = _issued()
    return _Pending(x, dist.all_reduce(x, op=op, async_op=True), None, tag, e)


def all_reduce_(x: torch.Tensor, op=dist.ReduceOp.SUM) -> torch.Tensor:
    if _native(x) and x.is_contiguous():
        return _NATIVE.all_reduce_(x, _native_op(op))
    if _host_staged(x):
        h = x.cpu()
        dist.all_reduce(h, op=op)
        x.copy_(h)
    else:
        dist.all_reduce(x, op=op)
    return x


def broadcast_(x: torch.Tensor, src: int) -> torch.Tensor:
    if _native(x) and x.is_contiguous():
        return _NATIVE.broadcast_(x, src)
    if _host_staged(x):
        h = x.cpu()
        dist.broadcast(h, src=src)
        x.copy_(h)
    else:
        dist.broadcast(x, src=src)
    return x


def all_to_all_rows(x: torch.Tensor) -> torch.Tensor:
    """Chunk r of `x` (equal chunks along dim 0) goes to rank r."""
    if _native(x):
        return _NATIVE.all_to_all(x)
    if _host_staged(x):
        return all_to_all_rows(x.cpu()).to(x.device)
    out = torch.empty_like(x)
    dist.all_to_all_single(out, x.contiguous())
    return out


def all_to_all_rows_start(x: torch.Tensor, tag: str = "all_to_all") -> _Pending:
    if _native(x):
        return _native_start(_NATIVE.all_to_all, x, None, tag=tag)
    if not _rccl_async(x):
        e = _issued()
        return _Pending(all_to_all_rows(x) if dist.get_world_size() > 1 else x, tag=tag, issued=e)
    x = x.contiguous()
    out = torch.empty_like(x)
    e = _issued()
    return _Pending(out, dist.all_to_all_single(out, x, async_op=True), x, tag, e)


# ----------------------------------------------------------------- product backend
class HipBackend:
    """All arithmetic on libtt_hotpath.so.  Raises if the library or the GPU is missing."""

    def __init__(self, device: torch.device):
        from . import _native, ops
        if device.type != "cuda":
            raise RuntimeError("HipBackend needs an MI355X device; there is no CPU path")
        self.N, self.ops = _native, ops
        self.lib = _native.load()
        self.device = device
        self._sides = {}
        self._du_unit = None
        self._kept = self._kept16 = None
        self._side_done, self._side_hold = None, []
        self.keep_logits = False  # set by the trainer when the logits dominate the step
        self._side_stream = None
        self._sweep_done = None

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def gather_owned(self, table: torch.Tensor, local: torch.Tensor, n_local: int) -> torch.Tensor:
        """out[i] = table[local[i]] for local[i] < n_local, a zero row for the sentinel."""
        out = self.empty(local.numel(), table.shape[1])
        N = self.N
        if n_local <= 0:  # a rank that owns no row of this table (fewer rows than ranks): every id is the sentinel
            return out.zero_()
        N.check(self.lib.tt_gather_rows(table.data_ptr(), n_local, table.shape[1], local.data_ptr(), local.numel(),
                                        out.data_ptr(), table.shape[1], None, N.stream()), "tt_gather_rows")
        return out

    def gather_rows(self, src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        """out[i] = src[idx[i]], a zero row where idx[i] is outside [0, len(src)) (padding slots)."""
        return self.gather_owned(src, idx, src.shape[0])

    # ---- owner routing (csrc/route.hip)
    def route_plan(self, ids: torch.Tensor, n_rows: int, rows_per_rank: int, world: int, max_out: torch.Tensor):
        """Count this rank's ids per owner (one stable counting pass, no sort); the largest bucket is
        atomicMax'ed into the int32 scalar view `max_out`.  Ids outside [0, n_rows) raise the device-side
        out-of-range flag (IndexError at the next poll, like the single-GPU lookups)."""
        N, lib = self.N, self.lib
        n = ids.numel()
        ws = torch.empty(lib.tt_route_workspace_bytes(n, world), dtype=torch.uint8, device=self.device)
        counts = torch.empty(world, dtype=torch.int32, device=self.device)
        N.check(lib.tt_route_count(ids.data_ptr(), n, n_rows, rows_per_rank, world, counts.data_ptr(), max_out.data_ptr(),
                                   N.oob.flag(self.device).data_ptr(), ws.data_ptr(), ws.numel(), N.stream()),
                "tt_route_count")
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

    def poll(self) -> None:
        self.N.oob.poll(self.device)

    def value_weights(self, labels: torch.Tensor, uvw: torch.Tensor):
        """-> (nuv [B] = clamp(labels . uvw, 1e-6), its maximum as a 1-element tensor) -- tt_value_weights."""
        labels = labels.contiguous()
        B, T = labels.shape
        nuv, mx = self.empty(B), self.empty(1)
        self.N.check(self.lib.tt_value_weights(labels.data_ptr(), B, T, uvw.contiguous().data_ptr(), nuv.data_ptr(), mx.data_ptr(),
                                               self.N.stream()), "tt_value_weights")
        return nuv, mx

    def weighted_loss_global(self, nuv, gmax, ce, denom: float):
        """-> (coef [B] = nuv / gmax / denom, loss scalar = sum(ce * nuv / gmax) / denom) -- tt_weighted_loss_global."""
        B = nuv.shape[0]
        coef, loss = self.empty(B), torch.empty((), dtype=torch.float32, device=self.device)
        self.N.check(self.lib.tt_weighted_loss_global(nuv.data_ptr(), gmax.data_ptr(), ce.contiguous().data_ptr(), B, float(denom),
                                                      coef.data_ptr(), loss.data_ptr(), self.N.stream()), "tt_weighted_loss_global")
        return coef, loss

    def tower_fwd(self, emb: torch.Tensor, feats: torch.Tensor, p: Sequence[torch.Tensor],
                  extra: Optional[torch.Tensor] = None):
        """out = [emb | MLP(feats) (| extra)] W3^T + b3 without building the concatenation: one
        product per column block of W3, accumulating.  `extra` [B, E] is the history summary of the
        history model's user tower (ref:src/two_tower_with_user_history_encoder.py:110-121).
        Returns (h, f, out)."""
        W1, b1, W2, b2, W3, b3 = p
        ops, N = self.ops, self.N
        B, F = feats.shape
        Dm, Hd = W2.shape
        E = 0 if extra is None else extra.shape[1]
        De = W3.shape[1] - Dm - E
        if (emb.is_contiguous() and (extra is None or (extra.stride(1) == 1 and extra.stride(0) % 4 == 0 and extra.data_ptr() % 16 == 0))
                and ops.fused_tower_supported(emb, feats, W1, W2, W3, extra_width=E)):
            # one launch (csrc/tower.hip): the routed rows play the table, looked up by position; returns
            # (h, tin, out) -- tower_bwd recognises the [B, 2D] tower input in the `f` slot
            pos = ops.ActiveStash.positions_for(self.device, B)
            h, tin, out = self.empty(B, Hd), self.empty(B, 2 * De), self.empty(B, W3.shape[0])
            feats = feats.contiguous()
            N.check(self.lib.tt_tower_fwd_x(emb.data_ptr(), B, pos.data_ptr(), feats.data_ptr(), feats.stride(0), B, De, F, Hd,
                                            W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr(),
                                            W3.shape[0], N.ptr(extra), extra.stride(0) if E else 0, E, out.data_ptr(),
                                            out.stride(0), h.data_ptr(), tin.data_ptr(), None, N.stream()), "tt_tower_fwd_x")
            return h, tin, out
        h = self.empty(B, Hd)
        ops.gemm(N.TT_GEMM_NT, feats, W1, h, B, Hd, F, bias=b1, epilogue=N.TT_EPI_RELU)
        f = self.empty(B, Dm)
        ops.gemm(N.TT_GEMM_NT, h, W2, f, B, Dm, Hd, bias=b2)
        out = self.empty(B, W3.shape[0])
        ops.gemm(N.TT_GEMM_NT, emb, W3[:, :De], out, B, W3.shape[0], De, bias=b3)
        ops.gemm(N.TT_GEMM_NT, f, W3[:, De:De + Dm], out, B, W3.shape[0], Dm, accumulate=True)
        if extra is not None:
            ops.gemm(N.TT_GEMM_NT, extra, W3[:, De + Dm:], out, B, W3.shape[0], E, accumulate=True)
        return h, f, out

    def tower_bwd(self, d_out, emb, h, f, feats, p, g, extra: Optional[torch.Tensor] = None):
        """Writes the six parameter gradients into `g` (views of the flat gradient buffer) and
        returns (gradient of the id-embedding rows [B, D_emb], gradient of `extra` or None)."""
        W1, b1, W2, b2, W3, b3 = p
        gW1, gb1, gW2, gb2, gW3, gb3 = g
        ops, N = self.ops, self.N
        B, F = feats.shape
        Dm, Hd = W2.shape
        Do, Din = W3.shape
        E = 0 if extra is None else extra.shape[1]
        De = Din - Dm - E
        if f.shape[1] == Din - E and f.shape[1] != Dm:  # the fused forward ran: `f` is the tower input [emb | MLP]
            d_out = d_out.contiguous()
            d_emb, d_f, dh = self.empty(B, De), self.empty(B, Dm), self.empty(B, Hd)
            d_extra = self.empty(B, E) if E else None
            N.check(self.lib.tt_tower_bwd_data_x(d_out.data_ptr(), d_out.stride(0), B, De, Hd, W2.data_ptr(), W3.data_ptr(),
                                                 h.data_ptr(), d_emb.data_ptr(), De, d_f.data_ptr(), dh.data_ptr(),
                                                 N.ptr(d_extra), E, E, N.stream()), "tt_tower_bwd_data_x")
            feats_c = feats.contiguous()
            if _WGRAD_ASIDE and self.device.type == "cuda":
                # the six weight gradients feed nothing but the dense all-reduce at the end of the step: they run on the third
                # stream underneath the other tower's backward and the row-gradient exchanges, joined by join_side()
                aux = N.aux_stream(self.device)
                ev = torch.cuda.Event()
                ev.record()
                aux.wait_event(ev)
                with torch.cuda.stream(aux):
                    ops.tower_weight_grads(d_out, f, d_f, h, dh, feats_c, out=(gW1, gb1, gW2, gb2, gW3, gb3), extra=extra)
                    self._side_done = torch.cuda.Event()
                    self._side_done.record()
                self._side_hold += [d_out, f, d_f, h, dh, feats_c, extra]  # (allocated on the main stream: alive until the join)
            else:
                ops.tower_weight_grads(d_out, f, d_f, h, dh, feats_c, out=(gW1, gb1, gW2, gb2, gW3, gb3), extra=extra)
            return d_emb, d_extra
        ops.gemm_tn_colsum(d_out, emb, gW3[:, :De], db=gb3)
        ops.gemm(N.TT_GEMM_TN, d_out, f, gW3[:, De:De + Dm], Do, Dm, B)
        d_emb = self.empty(B, De)
        ops.gemm(N.TT_GEMM_NN, d_out, W3[:, :De], d_emb, B, De, Do)
        d_f = self.empty(B, Dm)
        ops.gemm(N.TT_GEMM_NN, d_out, W3[:, De:De + Dm], d_f, B, Dm, Do)
        d_extra = None
        if extra is not None:
            ops.gemm(N.TT_GEMM_TN, d_out, extra, gW3[:, De + Dm:], Do, E, B)
            d_extra = self.empty(B, E)
            ops.gemm(N.TT_GEMM_NN, d_out, W3[:, De + Dm:], d_extra, B, E, Do)
        ops.gemm_tn_colsum(d_f, h, gW2, db=gb2)
        dh = self.empty(B, Hd)
        ops.gemm(N.TT_GEMM_NN, d_f, W2, dh, B, Hd, Dm, epilogue=N.TT_EPI_RELU_MASK, aux=h)
        ops.gemm_tn_colsum(dh, feats, gW1, db=gb1)
        return d_emb, d_extra

    def join_side(self) -> None:
        """The main stream waits for the tower weight gradients queued on the third stream (tower_bwd)."""
        if self._side_done is not None:
            torch.cuda.current_stream().wait_event(self._side_done)
            self._side_done = None
        self._side_hold = []

    # history encoder (ref:src/user_history_encoder.py:80-121) through the product's autograd function
    def encoder_fwd(self, x: torch.Tensor, pe: Optional[torch.Tensor], heads: int, layer_params: Sequence[torch.Tensor]):
        """x [B, H, D] embedded history -> (summary [B, 2, D], saved state for encoder_bwd)."""
        xin = x.detach().requires_grad_(True)
        ps = [t.detach().requires_grad_(True) for t in layer_params]
        with torch.enable_grad():
            out = self.ops.HistoryEncoder.apply(xin, None, pe, heads, *ps)
        return out.detach(), (out, xin, ps)

    def encoder_bwd(self, saved, d_summary: torch.Tensor, grad_views: Sequence[torch.Tensor]) -> torch.Tensor:
        """-> d_x [B*H, D]; the layer-parameter gradients are written into `grad_views`."""
        out, xin, ps = saved
        grads = torch.autograd.grad(out, [xin] + ps, d_summary.contiguous())
        for view, gr in zip(grad_views, grads[1:]):
            view.copy_(gr)
        return grads[0].reshape(-1, xin.shape[-1])

    def ce_fwd(self, U, I_all, off):
        ops, N, lib = self.ops, self.N, self.lib
        M, D = U.shape
        Nn = I_all.shape[0]
        lse, ce = self.empty(M), self.empty(M)
        wsp, wsn = ops._ws(self.device, lib.tt_inbatch_ce_workspace_bytes(M, Nn, D))
        # forward fused with the user-side gradient (kept for ce_bwd): with global negatives the
        # logits are the step at 8 GPUs, and this removes one of their five passes
        self._du_unit = self.empty(M, D)
        self._kept = self._kept16 = None
        if _CE_F16X2 and lib.tt_ce16_supported(M, Nn, D) and U.is_contiguous() and I_all.is_contiguous():
            # EXPLORATORY (TT_CE_F16X2=1): the same pair on the 16-bit matrix pipe, every product as three fp16 MFMA products
            # of two-term splits -- fp32-grade results (csrc/ce_f16x2.hip), about half the time of the fp32-MFMA pair
            w16p, w16n = ops._ws(self.device, lib.tt_ce16_workspace_bytes(M, Nn, D), "ce16")
            if _CE16_KEEP:
                self._kept16 = torch.empty(M * Nn, dtype=torch.float32, device=self.device)
                zp, zn = self._kept16.data_ptr(), M * Nn * 4
            else:
                # no logits buffer: on the fp16 pipe forming a tile again is cheaper than 8 bytes of HBM traffic per logit,
                # and half of this step's HBM traffic was the logits going out and coming back.  The backward reuses the images this call
                # leaves in the "ce16" workspace slot -- nothing else in the step touches that slot.
                self._kept16, zp, zn = (w16p, w16n), None, 0
            N.check(lib.tt_ce16_fwd_du_keep(U.data_ptr(), D, I_all.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), ce.data_ptr(),
                                            self._du_unit.data_ptr(), D, zp, zn, w16p, w16n, N.stream()), "tt_ce16_fwd_du_keep")
            return ce, lse
        if self.keep_logits and ops.kept_logits_supported(U, I_all):
            # wide negative sets: the logits are written out once and read back by the item-side
            # backward instead of being recomputed (3 instead of 4 logit-sized products per step)
            zn = lib.tt_inbatch_ce_logits_bytes(M, Nn)
            self._kept = torch.empty(zn, dtype=torch.uint8, device=self.device)
            N.check(lib.tt_inbatch_ce_fwd_du_keep(U.data_ptr(), D, I_all.data_ptr(), D, M, Nn, D, off, lse.data_ptr(),
                                                  ce.data_ptr(), self._du_unit.data_ptr(), D, self._kept.data_ptr(), zn,
                                                  wsp, wsn, N.stream()), "tt_inbatch_ce_fwd_du_keep")
            return ce, lse
        N.check(lib.tt_inbatch_ce_fwd_du(U.data_ptr(), D, I_all.data_ptr(), D, M, Nn, D, off, lse.data_ptr(),
                                         ce.data_ptr(), self._du_unit.data_ptr(), D, wsp, wsn, N.stream()),
                "tt_inbatch_ce_fwd_du")
        return ce, lse

    def ce_du(self, coef):
        """dU = dL/dce (.) du_unit: the user-side gradient, from the forward's unit form alone (no logits kernel)."""
        N, lib = self.N, self.lib
        M, D = self._du_unit.shape
        dU = self.empty(M, D)
        N.check(lib.tt_scale_rows(self._du_unit.data_ptr(), D, coef.data_ptr(), M, D, dU.data_ptr(), D, N.stream()), "tt_scale_rows")
        self._du_unit = None
        return dU

    def ce_bwd(self, U, I_all, off, lse, coef, want_du: bool = True):
        ops, N, lib = self.ops, self.N, self.lib
        M, D = U.shape
        Nn = I_all.shape[0]
        dI = self.empty(Nn, D)
        dU = self.ce_du(coef) if want_du else None
        if self._kept16 is not None:
            w16p, w16n = ops._ws(self.device, lib.tt_ce16_workspace_bytes(M, Nn, D), "ce16")
            if isinstance(self._kept16, tuple):
                same = self._kept16 == (w16p, w16n)  # (the slot was not re-allocated in between: the forward's images are there)
                N.check(lib.tt_ce16_bwd_recompute(U.data_ptr(), D, I_all.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), coef.data_ptr(),
                                                  dI.data_ptr(), D, w16p, w16n, 1 if same else 0, N.stream()), "tt_ce16_bwd_recompute")
            else:
                N.check(lib.tt_ce16_bwd_kept(U.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), coef.data_ptr(), self._kept16.data_ptr(),
                                             M * Nn * 4, dI.data_ptr(), D, w16p, w16n, N.stream()), "tt_ce16_bwd_kept")
            self._kept16 = None
            return dU, dI
        wsp, wsn = ops._ws(self.device, lib.tt_inbatch_ce_workspace_bytes(M, Nn, D))
        if self._kept is not None:
            N.check(lib.tt_inbatch_ce_bwd_kept(U.data_ptr(), D, M, Nn, D, off, lse.data_ptr(), coef.data_ptr(),
                                               self._kept.data_ptr(), self._kept.numel(), dI.data_ptr(), D, wsp, wsn,
                                               N.stream()), "tt_inbatch_ce_bwd_kept")
            self._kept = None
            return dU, dI
        N.check(lib.tt_inbatch_ce_bwd(U.data_ptr(), D, I_all.data_ptr(), D, M, Nn, D, off, lse.data_ptr(),
                                      coef.data_ptr(), None, D, dI.data_ptr(), D, wsp, wsn, N.stream()),
                "tt_inbatch_ce_bwd")
        return dU, dI

    def new_hyper(self, lr, betas, eps):
        return torch.tensor([lr, betas[0], betas[1], eps, 0, 0, 0, 0], dtype=torch.float64, device=self.device)

    def adam_advance(self, hyper):
        self.N.check(self.lib.tt_adam_advance(hyper.data_ptr(), self.N.stream()), "tt_adam_advance")

    # the table step in three phases (stash -> sweep on a side stream -> finish), see optim.py
    def adam_table_begin(self, W, M, V, n_local: int, local_ids: torch.Tensor):
        """`local_ids` may contain the sentinel n_local: the plan sorts it last and the Adam
        kernels skip its run."""
        ops, N, lib = self.ops, self.N, self.lib
        n_rows, dim = n_local, W.shape[1]
        if n_rows <= 0:  # this rank owns no row of the table (fewer rows than ranks): nothing to park, nothing to finish
            return None, None, 0
        # The old rows are parked by OCCURRENCE (slot i = occurrence i: needs nothing but the ids), so the sweep can start
        # without the sorted plan; the sort itself -- one 118 KB-LDS workgroup for <= 10 K ids, ~15 launches beyond -- runs
        # on a third stream underneath the towers and is waited for by adam_table_finish only (round 3: 85 us off the top
        # of the W = 8 step; it was plan -> stash -> plan -> stash in line, in front of everything).
        plan = ops.RowPlan([local_ids], n_rows + 1, slot=f"plan{W.data_ptr()}", defer=True)
        key = W.data_ptr()
        need = lib.tt_adam_table_workspace_bytes(plan.n, dim)
        side = self._sides.get(key)
        if side is None or side.numel() < need:
            side = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.device)
            self._sides[key] = side
        N.check(lib.tt_adam_table_stash_ids(W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, dim, plan.ids.data_ptr(), plan.n,
                                            side.data_ptr(), side.numel(), N.stream()), "tt_adam_table_stash_ids")
        main, aux = torch.cuda.current_stream(), N.aux_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)  # the localised ids exist
        aux.wait_event(ready)
        with torch.cuda.stream(aux):
            plan.build()
            done = torch.cuda.Event()
            done.record(aux)
        return plan, side, n_rows, done

    def adam_begin_tables(self, hyper, tables):
        """adam_advance + adam_table_begin for every (W, M, V, n_local, local_ids) as ONE launch (tt_adam_begin_ids); the
        row plans sort on the third stream as in adam_table_begin.  -> the per-table states."""
        ops, N, lib = self.ops, self.N, self.lib
        jobs = (N.AdamStashJob * max(len(tables), 1))()
        states, n_jobs = [], 0
        for W, M, V, n_local, local_ids in tables:
            if n_local <= 0:
                states.append((None, None, 0, None))
                continue
            dim = W.shape[1]
            plan = ops.RowPlan([local_ids], n_local + 1, slot=f"plan{W.data_ptr()}", defer=True)
            need = lib.tt_adam_table_workspace_bytes(plan.n, dim)
            side = self._sides.get(W.data_ptr())
            if side is None or side.numel() < need:
                side = torch.empty(int(need * 1.25) + 256, dtype=torch.uint8, device=self.device)
                self._sides[W.data_ptr()] = side
            j = jobs[n_jobs]
            j.W, j.M, j.V, j.n_rows, j.dim = W.data_ptr(), M.data_ptr(), V.data_ptr(), n_local, dim
            j.ids, j.n_ids, j.side, j.side_bytes = plan.ids.data_ptr(), plan.n, side.data_ptr(), side.numel()
            n_jobs += 1
            states.append([plan, side, n_local, None])
        N.check(lib.tt_adam_begin_ids(hyper.data_ptr(), None, 0, jobs, n_jobs, N.stream()), "tt_adam_begin_ids")
        main, aux = torch.cuda.current_stream(), N.aux_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        aux.wait_event(ready)
        with torch.cuda.stream(aux):
            for st in states:
                if st[0] is not None:
                    st[0].build()
            done = torch.cuda.Event()
            done.record(aux)
        return [tuple(st[:3]) + (done,) if st[0] is not None else st for st in states]

    def adam_finish_tables(self, hyper, tables) -> None:
        """adam_table_finish for every (W, M, V, state, grad_rows) as ONE launch (tt_adam_tables_finish)."""
        N, lib = self.N, self.lib
        live = [t for t in tables if t[3][0] is not None]
        if not live:
            return
        jobs = (N.AdamFinishJob * len(live))()
        waited = set()
        for i, (W, M, V, state, grad_rows) in enumerate(live):
            plan, side, n_rows, done = state
            if id(done) not in waited:
                torch.cuda.current_stream().wait_event(done)
                waited.add(id(done))
            plan.attach([grad_rows])
            j = jobs[i]
            j.W, j.M, j.V, j.n_rows, j.dim = W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, W.shape[1]
            j.src, j.n_ids = C.pointer(plan.sources), plan.n
            j.sorted_ids, j.perm, j.seg_begin, j.n_unique = (plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                                             plan.seg_begin.data_ptr(), plan.n_unique.data_ptr())
            j.side, j.side_bytes = side.data_ptr(), side.numel()
        N.check(lib.tt_adam_tables_finish(jobs, len(live), hyper.data_ptr(), N.stream()), "tt_adam_tables_finish")

    def sweep_async(self, tables, hyper, n_wgs: int = 0):
        """Zero-gradient sweep of every (W, M, V) on the side stream, after everything queued so
        far on the main stream (the lookups and the stashes)."""
        N, lib = self.N, self.lib
        if self._side_stream is None:
            self._side_stream = self.N.low_priority_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self._side_stream.wait_event(ready)
        live = [(W, M, V, n_local) for W, M, V, n_local in tables if n_local > 0]
        if live:  # one launch for both row blocks
            descs = (N.AdamTensor * len(live))()
            for i, (W, M, V, n_local) in enumerate(live):
                descs[i].p, descs[i].g, descs[i].m, descs[i].v = W.data_ptr(), None, M.data_ptr(), V.data_ptr()
                descs[i].n = n_local * W.shape[1]
            N.check(lib.tt_adam_tables_sweep(descs, len(live), hyper.data_ptr(), n_wgs, self._side_stream.cuda_stream),
                    "tt_adam_tables_sweep")
        self._sweep_done = torch.cuda.Event()
        self._sweep_done.record(self._side_stream)

    def sweep_wait(self):
        torch.cuda.current_stream().wait_event(self._sweep_done)

    def adam_table_finish(self, W, M, V, hyper, state, grad_rows: torch.Tensor):
        N, lib = self.N, self.lib
        plan, side, n_rows = state[:3]
        if plan is None:  # empty row block
            return
        torch.cuda.current_stream().wait_event(state[3])  # the sorted plan (third stream, see adam_table_begin)
        plan.attach([grad_rows])
        N.check(lib.tt_adam_table_finish(W.data_ptr(), M.data_ptr(), V.data_ptr(), n_rows, W.shape[1],
                                         hyper.data_ptr(), C.byref(plan.sources), plan.n, plan.sorted_ids.data_ptr(),
                                         plan.perm.data_ptr(), plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(),
                                         side.data_ptr(), side.numel(), N.stream()), "tt_adam_table_finish")

    def mips_topk(self, query, corpus, k):
        return self.ops.mips_topk(query, corpus, k)

    def mips_merge(self, scores, idx, k):
        return self.ops.mips_merge(scores, idx, k)

    def adam_dense(self, p, g, m, v, hyper):
        N = self.N
        d = (N.AdamTensor * 1)()
        d[0].p, d[0].g, d[0].m, d[0].v, d[0].n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        N.check(self.lib.tt_adam_dense(d, 1, hyper.data_ptr(), N.stream()), "tt_adam_dense")


# ----------------------------------------------------------------- the sharded trainer
class ShardedTable:
    """Contiguous row block [lo, hi) of a [n_rows, dim] table, with its Adam moments."""

    def __init__(self, n_rows: int, dim: int, device, generator_seed: int):
        W, r = dist.get_world_size(), dist.get_rank()
        self.n_rows, self.dim = n_rows, dim
        self.rows_per_rank = (n_rows + W - 1) // W
        self.lo = min(r * self.rows_per_rank, n_rows)
        self.hi = min(self.lo + self.rows_per_rank, n_rows)
        n_local = max(self.hi - self.lo, 1)
        gen = torch.Generator(device=device).manual_seed(generator_seed + 7919 * r)
        self.weight = torch.randn(n_local, dim, generator=gen, device=device)  # nn.Embedding init: N(0,1)
        self.m = torch.zeros_like(self.weight)
        self.v = torch.zeros_like(self.weight)


class Lookup:
    """Every rank's ids of one lookup, seen from this rank's block: `local` holds the row
    offset inside the block for ids this rank owns and the sentinel `n_local` (one past the
    block) for everybody else's."""

    def __init__(self, ids: torch.Tensor, table: ShardedTable):
        ids_all = all_gather_rows(ids)
        local = ids_all - table.lo
        owned = (local >= 0) & (local < (table.hi - table.lo))
        self.n_local = table.hi - table.lo
        self.local = torch.where(owned, local, torch.full_like(local, self.n_local))


class _PlannedRoutes:
    """The routing of one batch's lookups as far as it can be prepared without knowing `cap`: per lookup
    the sorted ids + bucket starts, and the all-reduced bucket maxima on their way to the host."""

    __slots__ = ("key", "planned", "counts_host", "event", "keep")

    def __init__(self, key, planned, counts_host, event, keep):
        self.key, self.planned, self.counts_host, self.event, self.keep = key, planned, counts_host, event, keep

    def caps(self) -> List[int]:
        if self.event is not None:
            self.event.synchronize()  # planned a step ahead: long complete, no wait
        return [max(64, (int(c) + 63) // 64 * 64) for c in self.counts_host.tolist()]


class _RoutedLookup:
    """One lookup in flight: requester side (slot_of, src_of) and owner side (local, n_local)."""

    __slots__ = ("table", "cap", "slot_of", "src_of", "ids_p", "rows_p", "local", "n_local")

    def __init__(self, table, cap, slot_of, src_of, ids_p):
        self.table, self.cap, self.slot_of, self.src_of, self.ids_p = table, cap, slot_of, src_of, ids_p
        self.rows_p, self.local = None, None
        self.n_local = table.hi - table.lo


class _ScheduleScan:
    """Pick the fastest of a few step schedules by MEASURING them (the sharded step's counterpart of
    DenseExactAdam._tune_sweep): full steps are timed with events recorded on the main stream, queried but never
    waited on.  Each candidate runs one block of `block` consecutive steps, enqueued open loop (the host may be many
    steps ahead of the GPU); the first step of a block overlaps the previous candidate's tail and is not counted; once
    the last block's events are in, the candidate with the smallest step time is kept.  Results never depend on the
    choice: every candidate is the same arithmetic in the same order."""

    def __init__(self, candidates, block: int = 6, skip_first: int = 4, new_event=None, group_max=None):
        self._new_event = new_event or (lambda: torch.cuda.Event(enable_timing=True))  # (tests: stub events)
        # group_max(list of floats) -> element-wise MAX over the ranks.  With it the choice is made ONCE FOR THE GROUP, at a
        # step number every rank reaches (`decide_at`): the steps are synchronised by the collectives, so the fastest
        # candidate is a property of the group, and ranks that locked different schedules on local timing noise would
        # drag each other.  Without it (one rank, tests) the local measurements decide as soon as they are in.
        self._group_max = group_max
        self.cands = list(candidates)
        self.plan = [c for c in self.cands for _ in range(block)]
        self.block, self.skip_first = block, skip_first
        self.n = 0
        self.pending: list = []
        self.obs = {c: [] for c in self.cands}
        self.best = None
        self._cur = None
        self.decide_at = skip_first + len(self.plan) + 4  # group decision: a few steps after the last block was enqueued
        self.decided_by = None

    def begin(self):
        """-> the candidate this step runs with; call end() when the step has been enqueued."""
        if self.best is not None:
            return self.best
        self._drain()
        if self.best is not None:
            return self.best
        if self._group_max is not None and self.n >= self.decide_at:
            self._decide_group()
            return self.best
        k = self.n - self.skip_first
        self.n += 1
        if k >= len(self.plan):  # every block has been enqueued: first candidate until the last measurements are in
            self._cur = None
            return self.cands[0]
        cand = self.cands[0] if k < 0 else self.plan[k]
        ev = self._new_event()
        ev.record()
        self._cur = [ev, None, cand, k >= 0 and k % self.block != 0]
        return cand

    def end(self) -> None:
        if self.best is not None or self._cur is None:
            return
        ev = self._new_event()
        ev.record()
        self._cur[1] = ev
        self.pending.append(self._cur)
        self._cur = None

    def _drain(self) -> None:
        while self.pending and self.pending[0][1].query():
            a, b, cand, counted = self.pending.pop(0)
            if counted:
                self.obs[cand].append(a.elapsed_time(b))
        if self._group_max is None and self.n >= self.skip_first + len(self.plan) and not self.pending:
            ms = {c: min(v) for c, v in self.obs.items() if len(v) >= 2}
            self.best = min(ms, key=ms.get) if ms else self.cands[0]
            self.decided_by = "local measurements"
            if os.environ.get("TT_TUNE_DEBUG"):
                import sys
                print(f"[tt] sharded step schedule (sweep start, workgroups): {ms} -> {self.best}", file=sys.stderr)

    def _decide_group(self) -> None:
        """Every rank calls this at the same step: wait for the own measurements (one host wait, once per run), take the
        MAX over the ranks of each candidate's best time -- the group moves at the pace of its slowest rank -- and keep
        the candidate whose worst rank is fastest.  Identical inputs on every rank -> identical choice."""
        for a, b, cand, counted in self.pending:
            b.synchronize()
            if counted:
                self.obs[cand].append(a.elapsed_time(b))
        self.pending = []
        local = [min(self.obs[c]) if len(self.obs[c]) >= 2 else 1.0e9 for c in self.cands]
        worst = list(self._group_max(local))
        self.best = self.cands[min(range(len(self.cands)), key=lambda i: (worst[i], i))]
        self.decided_by = "group (max over ranks of each candidate's best step time)"
        if os.environ.get("TT_TUNE_DEBUG"):
            import sys
            print(f"[tt] sharded step schedule: local {local}, group {worst} -> {self.best}", file=sys.stderr)


class ShardedTrainer:
    """TwoTowerBaseRetrieval train step (ref:src/two_tower_base_retrieval.py:349-394 +
    ref:train/train.py:112-125) on W row-sharded ranks.  `cfg`: n_users, n_items, D, F, B; with
    `model="hist"` and `H` the TwoTowerWithUserHistoryEncoder step
    (ref:src/two_tower_with_user_history_encoder.py:85-122): the B*H history rows travel through
    the same owner-gather / reduce-scatter exchange as the id rows, the encoder is replicated."""

    def __init__(self, cfg: Dict, device: torch.device, negatives: str = "global", backend=None,
                 lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, user_value_weights=(1.0,), seed: int = 0,
                 dense_init: Optional[Dict[str, torch.Tensor]] = None, routing: Optional[str] = None,
                 transport: Optional[str] = None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedTrainer needs torch.distributed to be initialised")
        self.cfg, self.device, self.negatives = dict(cfg), device, negatives
        # which library moves the bytes: torch.distributed's process group, or the C ABI's tt_comm_* (RCCL bound by
        # libtt_hotpath.so itself).  Same collectives, same order, same results.
        self.transport = transport or os.environ.get("TT_COMM", "torch")
        if self.transport not in ("torch", "native"):
            raise ValueError("transport must be 'torch' or 'native'")
        if self.transport == "native" and device.type == "cuda" and (dist.get_world_size() > 1 or _force_async()):
            if _is_gloo():
                raise RuntimeError("transport='native' needs one GPU per rank (RCCL); this group runs over gloo")
            from .comm import NativeComm
            if _NATIVE is None:
                use_native_transport(NativeComm.from_torch_distributed(device))
        self.routing = routing or os.environ.get("TT_ROUTE", "alltoall")  # A/B switch
        if self.routing not in ("alltoall", "allgather"):
            raise ValueError("routing must be 'alltoall' or 'allgather'")
        self._planned_next: Optional[_PlannedRoutes] = None
        self.comm_bytes: Dict[str, int] = {}  # bytes this rank SENDS to other ranks per step, by exchange
        self.W, self.rank = dist.get_world_size(), dist.get_rank()
        self.be = backend if backend is not None else HipBackend(device)
        D, F = cfg["D"], cfg["F"]
        # history model (ref:src/two_tower_with_user_history_encoder.py): the encoder is replicated,
        # its input rows come out of the sharded ITEM table like every other lookup
        if cfg.get("model", "base") not in ("base", "hist"):
            raise ValueError(f"ShardedTrainer: model {cfg.get('model')!r} is not sharded (base and hist are)")
        self.hist = cfg.get("model", "base") == "hist"
        self.heads, self.layers = 4, 3  # hard-coded upstream (ref :64-70)
        self.users = ShardedTable(cfg["n_users"], D, device, seed + 1)
        self.items = ShardedTable(cfg["n_items"], D, device, seed + 2)
        self.uvw = torch.tensor(list(user_value_weights), dtype=torch.float32, device=device)
        # replicated dense parameters live in ONE flat buffer (one all_reduce, one Adam launch)
        shapes = []
        for side in ("user", "item"):
            tower_in = 4 * D if (self.hist and side == "user") else 2 * D
            shapes += [(f"{side}_features_arch.0.weight", (256, F)), (f"{side}_features_arch.0.bias", (256,)),
                       (f"{side}_features_arch.2.weight", (D, 256)), (f"{side}_features_arch.2.bias", (D,)),
                       (f"{side}_tower_arch.weight", (D, tower_in)), (f"{side}_tower_arch.bias", (D,))]
        self.encoder_keys: List[str] = []
        if self.hist:
            for l in range(self.layers):
                base = f"user_history_encoder.multihead_attn_layers.{l}."
                layer = [(base + "in_proj_weight", (3 * D, D)), (base + "in_proj_bias", (3 * D,)),
                         (base + "out_proj.weight", (D, D)), (base + "out_proj.bias", (D,))]
                shapes += layer
                self.encoder_keys += [k for k, _ in layer]
        total = sum(math.prod(s) for _, s in shapes)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        off = 0
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for name, shape in shapes:
            n = math.prod(shape)
            self.params[name] = self.flat_p[off:off + n].view(shape)
            self.grads[name] = self.flat_g[off:off + n].view(shape)
            if dense_init is not None:
                self.params[name].copy_(dense_init[name].to(device))
            elif len(shape) == 2:
                # nn.Linear default: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); MultiheadAttention's packed
                # in-projection: xavier-uniform.
                bound = (math.sqrt(6.0 / (shape[0] + shape[1])) if name.endswith("in_proj_weight")
                         else 1.0 / math.sqrt(shape[1]))
                self.params[name].copy_(((torch.rand(shape, generator=gen) * 2 - 1) * bound).to(device))
                fan_in = shape[1]
            elif "multihead_attn_layers" not in name:
                # nn.Linear biases draw from the same U(-1/sqrt(fan_in), ..) as their weight (the entry
                # just before); only the attention layers' in_proj / out_proj biases start at zero upstream
                bound = 1.0 / math.sqrt(fan_in)
                self.params[name].copy_(((torch.rand(shape, generator=gen) * 2 - 1) * bound).to(device))
            off += n
        broadcast_(self.flat_p, src=0)  # replicas must start bit-identical
        self.pe = None
        if self.hist:  # the reference's table, newest item first (ref:src/user_history_encoder.py:35-78)
            H = cfg["H"]
            table = torch.zeros(H, D)
            for pos in range(H):
                for c in range(D):
                    angle = pos / (10000 ** ((2 * c) / D))
                    table[pos, c] = math.sin(angle) if c % 2 == 0 else math.cos(angle)
            self.pe = table.flip([0]).contiguous().to(device)
        self.hyper = self.be.new_hyper(lr, betas, eps)
        # When does the zero-gradient sweep start?  A long one (thick row blocks) as early as possible.
        # A short one is better started with the logits kernels: it saturates HBM, which slows the
        # small latency-bound tower GEMMs 6x, but costs the MFMA-bound logits kernels little.
        sweep_ms = (self.users.weight.numel() + self.items.weight.numel()) * 24 / 6.0e9
        logits_ms = 8.0 * cfg["B"] * cfg["B"] * (self.W if negatives == "global" else 1) * cfg["D"] / 125.0e9
        # Round 2 (routed lookups, fused towers: the top of the step is no longer a string of tiny GEMMs): starting the
        # sweep at the top wins at every world size (emulated W = 8: 4.60 vs 5.12 ms, W = 4: 3.37 vs 3.59, W = 2: 3.41 vs
        # 3.83), and a THIN sweep (256 persistent workgroups instead of 768) that lasts longer takes less from the logits
        # kernels when they, not the sweep, are the step (W = 8: 4.49, W = 4: 3.18 ms).
        late = os.environ.get("TT_SWEEP_LATE")  # A/B switch
        self._sweep_late = late == "1"
        self._sweep_bwd = late == "2"  # start the sweep with the BACKWARD logits kernel
        self._sweep_wgs = 256 if sweep_ms < 0.75 * logits_ms else 0
        if os.environ.get("TT_SWEEP_WGS") is not None:  # A/B switch
            self._sweep_wgs = int(os.environ["TT_SWEEP_WGS"])
        # Round 3: when the logits kernels are the step (thin row blocks, several ranks' items per user) WHERE the sweep
        # starts and how wide it runs are worth 3-8 % of the step and the best pair is a narrow optimum (emulated W = 8:
        # top of the step / 256 workgroups 4.49 ms, with the backward logits kernel / 256: 4.38, / 320: 4.67, / 192:
        # 4.57) -- so it is measured on the first steps instead of guessed (see _ScheduleScan).  (start, workgroups),
        # start 0 = top of the step, 2 = with the backward logits kernel.
        self._scan = None
        if (late is None and os.environ.get("TT_SWEEP_WGS") is None and sweep_ms < 0.75 * logits_ms
                and getattr(self.be, "device", device).type == "cuda" and self.routing != "allgather"):
            self._scan = _ScheduleScan([(0, 256), (2, 256), (0, 0)], block=4, skip_first=3,  # 15 steps, then fixed
                                       group_max=self._group_max if self.W > 1 else None)
        # the same regime decides whether the forward keeps the logits for the backward (one product
        # fewer, M*N*4 B of HBM traffic each way more): worth it once the sweep no longer binds
        keep = os.environ.get("TT_CE_KEEP_LOGITS")
        if hasattr(self.be, "keep_logits"):
            self.be.keep_logits = (keep == "1") if keep is not None else sweep_ms < 0.75 * logits_ms
        self.last_loss = torch.zeros((), dtype=torch.float32, device=device)
        self._tower_stream = None
        self._hold = None

    def _group_max(self, values):
        t = torch.tensor(values, dtype=torch.float32, device=self.device)
        all_reduce_(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]

    def schedule_note(self):
        """Where the sweep starts / how wide it runs in this trainer's step, and who decided (bench.py prints it)."""
        names = {0: "top of the step", 1: "with the forward logits kernel", 2: "with the backward logits kernel"}
        start = 2 if self._sweep_bwd else 1 if self._sweep_late else 0
        return {"sweep_start": names[start], "sweep_workgroups": self._sweep_wgs or 768,
                "decided_by": (self._scan.decided_by or "scan still running") if self._scan is not None else "fixed by shape / environment"}

    def _tower_params(self, side):
        return [self.params[f"{side}_{k}"] for k in TOWER_KEYS]

    def _tower_grads(self, side):
        return [self.grads[f"{side}_{k}"] for k in TOWER_KEYS]

    def make_batches(self, n: int, seed: int = 1234) -> List[Tuple[torch.Tensor, ...]]:
        """Per-rank synthetic batches with the distributions of ref:train/train.py:47-65."""
        cfg = self.cfg
        gen = torch.Generator(device="cpu").manual_seed(seed + 1000 * self.rank)
        B, F = cfg["B"], cfg["F"]
        out = []
        for _ in range(n):
            b = (torch.randint(0, cfg["n_users"], (B,), generator=gen), torch.randn(B, F, generator=gen),
                 torch.randint(0, cfg["n_items"], (B, cfg.get("H", 1)), generator=gen),
                 torch.randint(0, cfg["n_items"], (B,), generator=gen), torch.randn(B, F, generator=gen),
                 torch.randint(0, 10, (B,), generator=gen), torch.randint(0, 2, (B, 1), generator=gen).float())
            out.append(tuple(t.to(self.device) for t in b))
        return out

    # ---- checkpoint adaptor: sharded blocks <-> the reference's state_dict
    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Full tables assembled on every rank, under the reference's Parameter names, so the
        result loads into TwoTowerBaseRetrieval (either implementation) unchanged."""
        out = {k: v.detach().clone() for k, v in self.params.items()}
        for name, table in (("user_id_embedding_arch.weight", self.users),
                            ("item_id_embedding_arch.weight", self.items)):
            per = table.rows_per_rank
            block = table.weight.new_zeros(per, table.dim)
            block[: table.hi - table.lo] = table.weight[: table.hi - table.lo]
            full = all_gather_rows(block) if self.W > 1 else block
            out[name] = full[: table.n_rows].clone()
        return out

    def load_state_dict(self, state: Dict[str, torch.Tensor]) -> None:
        """Scatter a reference-format state_dict into the row blocks / replicated buffer.
        Adam moments restart from zero (the reference never checkpoints its optimiser)."""
        with torch.no_grad():
            for k, v in self.params.items():
                v.copy_(state[k].to(v.device))
            for name, table in (("user_id_embedding_arch.weight", self.users),
                                ("item_id_embedding_arch.weight", self.items)):
                full = state[name]
                if full.shape != (table.n_rows, table.dim):
                    raise ValueError(f"{name}: expected {(table.n_rows, table.dim)}, got {tuple(full.shape)}")
                table.weight[: table.hi - table.lo].copy_(full[table.lo:table.hi].to(table.weight.device))
                table.m.zero_()
                table.v.zero_()

    def index_corpus(self, item_features_block: torch.Tensor) -> "ShardedMIPS":
        """Serve what was trained (upstream searches a random corpus,
        ref:src/baseline_mips_module.py:29-30): the item tower over THIS rank's rows of the catalogue -- item r is
        row r of the item table, `item_features_block` [hi - lo, II] the features of rows [items.lo, items.hi) --
        installed as this rank's block of a ShardedMIPS.  Its search() equals TwoTowerBaseRetrieval.index_corpus +
        forward on one device with the gathered state_dict()."""
        n = self.items.hi - self.items.lo
        feats = item_features_block.to(self.device, torch.float32)
        if feats.shape[0] != n:
            raise ValueError(f"item_features_block: expected {n} rows (this rank's item rows), got {feats.shape[0]}")
        if n > 0:
            with torch.no_grad():
                _, _, corpus = self.be.tower_fwd(self.items.weight[:n], feats, self._tower_params("item"))
        else:  # a rank that owns no item row
            corpus = torch.empty(0, self.cfg["D"], dtype=torch.float32, device=self.device)
        return ShardedMIPS(corpus, self.items.lo, backend=self.be)

    # ---- lookup through the owning ranks (fixed-size collectives, no host sync)
    def _lookup(self, table: ShardedTable, ids: torch.Tensor) -> Tuple[Lookup, "_Pending"]:
        """-> (routing, pending rows).  The reduce-scatter that delivers the rows is only STARTED here:
        the caller waits for it where the rows are first used, so the next lookup's gather and the
        other tower's GEMMs run underneath it."""
        lk = Lookup(ids, table)
        partial = self.be.gather_owned(table.weight, lk.local, lk.n_local)  # [W*B, D], zeros if not mine
        return lk, reduce_scatter_rows_start(partial)

    # ---- routed lookups (padded all-to-all)
    def _route_specs(self, batch):
        specs = [(self.users, batch[0].reshape(-1))]
        if self.hist:  # history rows first: the reference's lookup order on the item table
            specs.append((self.items, batch[2].reshape(-1)))
        specs.append((self.items, batch[3].reshape(-1)))
        return specs

    @staticmethod
    def _route_key(batch):
        # storage, size AND torch's in-place version counter: a static input buffer refilled with copy_() between
        # the announcement and the step keeps its address but not its version, and is planned again
        return tuple((t.data_ptr(), t.numel(), t._version) for t in (batch[0], batch[2], batch[3]))

    def plan_routes(self, batch) -> _PlannedRoutes:
        """Sort each lookup's ids by owner and start the all-reduce (MAX) of the bucket maxima + its copy to
        the host.  Called for the NEXT batch from inside `step`, so the answer is there long before the host
        needs it; a batch that was not announced is planned on the spot (one host synchronisation)."""
        specs = self._route_specs(batch)
        counts = torch.zeros(len(specs), dtype=torch.int32, device=self.device)
        planned, keep = [], []
        for k, (table, ids) in enumerate(specs):
            # a private copy: route_build ranks THESE ids against the bucket offsets counted here, whatever happens
            # to the caller's tensor in between (a writer torch does not see -- a prefetcher's raw memcpy -- would
            # otherwise put two ids into one send slot without tripping the overflow flag)
            ids = ids.to(torch.int64).contiguous().clone() if ids.dtype == torch.int64 and ids.is_contiguous() \
                else ids.to(torch.int64).contiguous()
            keep.append(ids)
            planned.append(self.be.route_plan(ids, table.n_rows, table.rows_per_rank, self.W, counts[k:k + 1]))
        if self.W > 1:
            all_reduce_start_(counts, op=dist.ReduceOp.MAX, tag="route_caps_allreduce").wait()
        if counts.is_cuda:
            host = torch.empty(len(specs), dtype=torch.int32).pin_memory()
            host.copy_(counts, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
            keep.append(counts)
        else:
            host, event = counts, None
        return _PlannedRoutes(self._route_key(batch), planned, host, event, keep)

    def step(self, batch, next_batch=None) -> torch.Tensor:
        """One train step on this rank's B rows.  `next_batch` (optional): the batch the NEXT call will
        bring -- its routes are planned underneath this step, which removes the step's only host wait."""
        if self.routing == "allgather":
            return self._step_allgather(batch)
        user_id, user_feat, hist_ids, item_id, item_feat, _pos, labels = batch
        be, W, D, B = self.be, self.W, self.cfg["D"], user_id.shape[0]
        if hasattr(be, "poll"):
            be.poll()  # an out-of-range id seen by an earlier step surfaces here as IndexError
        if self._scan is not None:
            start, self._sweep_wgs = self._scan.begin()
            self._sweep_late, self._sweep_bwd = False, start == 2
        routes = self._planned_next
        self._planned_next = None
        if routes is None or routes.key != self._route_key(batch):
            routes = self.plan_routes(batch)
        elif routes.event is not None:
            torch.cuda.current_stream().wait_event(routes.event)  # (planned on the third stream)
        aside = next_batch is not None and _PLAN_ASIDE and self.device.type == "cuda" and hasattr(be, "N")
        if aside:
            aux = be.N.aux_stream(self.device)
            ready = torch.cuda.Event()
            ready.record()  # whatever produced next_batch was queued on this stream before the call
            aux.wait_event(ready)
            with torch.cuda.stream(aux):
                self._planned_next = self.plan_routes(next_batch)
        caps = routes.caps()
        # 1. each owner is sent the ids it holds (cap slots per peer), and returns the rows in the same slots
        lks: List[_RoutedLookup] = []
        sent_ids = sent_rows = 0
        for (table, _ids), planned, cap in zip(self._route_specs(batch), routes.planned, caps):
            send_ids, slot_of, src_of = be.route_build(planned, table.rows_per_rank, W, cap)
            lks.append(_RoutedLookup(table, cap, slot_of, src_of, all_to_all_rows_start(send_ids, tag="lookup_ids_alltoall")))
            sent_ids += (W - 1) * cap * 8
            sent_rows += (W - 1) * cap * D * 4
        for lk in lks:
            lk.local = be.localize(lk.ids_p.wait(), lk.table.lo, lk.n_local)  # sentinel n_local for padding
            lk.rows_p = all_to_all_rows_start(be.gather_owned(lk.table.weight, lk.local, lk.n_local), tag="lookup_rows_alltoall")
        lk_u, lk_i = lks[0], lks[-1]
        lk_h = lks[1] if self.hist else None
        if next_batch is not None and not aside:
            self._planned_next = self.plan_routes(next_batch)
        item_local = torch.cat([lk_h.local, lk_i.local]) if self.hist else lk_i.local
        # the tables' old rows have been read: plan, park the looked-up rows, and start the
        # zero-gradient sweep on the side stream -- it overlaps everything up to step 6
        if hasattr(be, "adam_begin_tables"):  # advance + both stashes as one launch
            st_u, st_i = be.adam_begin_tables(self.hyper, [(self.users.weight, self.users.m, self.users.v, lk_u.n_local, lk_u.local),
                                                           (self.items.weight, self.items.m, self.items.v, lk_i.n_local, item_local)])
        else:
            be.adam_advance(self.hyper)
            st_u = be.adam_table_begin(self.users.weight, self.users.m, self.users.v, lk_u.n_local, lk_u.local)
            st_i = be.adam_table_begin(self.items.weight, self.items.m, self.items.v, lk_i.n_local, item_local)
        sweep = [(self.users.weight, self.users.m, self.users.v, lk_u.n_local),
                 (self.items.weight, self.items.m, self.items.v, lk_i.n_local)]
        if not self._sweep_late and not self._sweep_bwd:
            be.sweep_async(sweep, self.hyper, self._sweep_wgs)
        pu, pi = self._tower_params("user"), self._tower_params("item")
        summary, enc_saved = None, None
        u_emb = be.gather_rows(lk_u.rows_p.wait(), lk_u.slot_of)  # the other exchanges are still in flight
        if self.hist:
            enc_params = [self.params[k] for k in self.encoder_keys]
            h_rows = be.gather_rows(lk_h.rows_p.wait(), lk_h.slot_of)
            summary3, enc_saved = be.encoder_fwd(h_rows.view(B, -1, D), self.pe, self.heads, enc_params)
            summary = summary3.reshape(B, 2 * D)
        u_h, u_f, U = be.tower_fwd(u_emb, user_feat, pu, extra=summary)
        i_emb = be.gather_rows(lk_i.rows_p.wait(), lk_i.slot_of)
        i_h, i_f, I = be.tower_fwd(i_emb, item_feat, pi)
        # 2. logits against every rank's items
        glob = self.negatives == "global" and W > 1
        I_all = _timed_sync("item_emb_allgather", all_gather_rows, I) if glob else I
        off = self.rank * B if glob else 0
        if self._sweep_late:  # a short sweep hides under the logits kernels instead of the small tower GEMMs
            be.sweep_async(sweep, self.hyper, self._sweep_wgs)
        ce, lse = be.ce_fwd(U, I_all, off)
        loss, coef = self._weighted_loss(ce, labels, B, glob)
        if self._sweep_bwd:
            be.sweep_async(sweep, self.hyper, self._sweep_wgs)
        # 4. backward through the loss
        early = None
        if _UTOWER_EARLY and not self.hist and hasattr(be, "ce_du") and self.device.type == "cuda":
            # The user-side gradient dU exists as soon as the loss weights do (the forward kernel produced its unit form), so
            # the user tower's backward does not have to queue up behind the item-side logits kernel: it runs on a second
            # stream UNDERNEATH it (ce_bwd_kept's 202-register workgroups leave wave slots for the tower kernels).
            dU = be.ce_du(coef)
            if self._tower_stream is None:
                # the library's existing third stream (idle by now: it sorted the row plans at the top of the step), NOT a
                # new one: ROCm multiplexes HIP streams onto a handful of hardware queues in creation order, and a fifth
                # stream landed on the sweep's queue -- its kernels then waited for the persistent sweep, 6.2 ms per step
                self._tower_stream = be.N.aux_stream(self.device)
            ready = torch.cuda.Event()
            ready.record()
            self._tower_stream.wait_event(ready)
            with torch.cuda.stream(self._tower_stream):
                d_urows, d_summary = be.tower_bwd(dU, u_emb, u_h, u_f, user_feat, pu, self._tower_grads("user"), extra=summary)
                send_u = be.gather_rows(d_urows, lk_u.src_of)
                early = torch.cuda.Event()
                early.record()
            # (no Tensor.record_stream here: blocks marked that way are not reusable until the allocator has seen their
            # events, the pool then grows by hipMalloc every step and the HOST stalls -- 6.3 instead of 4.2 ms per step.
            # References held until the next step instead: by then this stream has waited for `early`.)
            self._hold = [dU, u_emb, u_h, u_f, user_feat, d_urows, d_summary, send_u]
            _dU2, dI_all = be.ce_bwd(U, I_all, off, lse, coef, want_du=False)
        else:
            dU, dI_all = be.ce_bwd(U, I_all, off, lse, coef)
        dI_p = reduce_scatter_rows_start(dI_all, tag="dI_reduce_scatter") if glob else _Pending(dI_all)  # travels under the user tower backward
        # 5. towers backward -> dense grads (flat buffer) + embedding-row grads; each row-gradient block
        # goes back through its lookup's slots as soon as it exists
        if early is not None:
            torch.cuda.current_stream().wait_event(early)
            g_u_p = all_to_all_rows_start(send_u, tag="rowgrad_alltoall")
        else:
            d_urows, d_summary = be.tower_bwd(dU, u_emb, u_h, u_f, user_feat, pu, self._tower_grads("user"), extra=summary)
            g_u_p = all_to_all_rows_start(be.gather_rows(d_urows, lk_u.src_of), tag="rowgrad_alltoall")  # aligned with lk_u.local
        g_h_p = None
        if self.hist:
            d_hrows = be.encoder_bwd(enc_saved, d_summary.view(B, 2, D), [self.grads[k] for k in self.encoder_keys])
            g_h_p = all_to_all_rows_start(be.gather_rows(d_hrows, lk_h.src_of), tag="rowgrad_alltoall")
        d_irows, _ = be.tower_bwd(dI_p.wait(), i_emb, i_h, i_f, item_feat, pi, self._tower_grads("item"))
        g_i_p = all_to_all_rows_start(be.gather_rows(d_irows, lk_i.src_of), tag="rowgrad_alltoall")
        if hasattr(be, "join_side"):
            be.join_side()
        flat_p = all_reduce_start_(self.flat_g, tag="dense_grad_allreduce")  # every dense gradient has been written by now
        g_u, g_i = g_u_p.wait(), g_i_p.wait()
        if self.hist:  # aligned with item_local = [history ids | item ids]
            g_i = torch.cat([g_h_p.wait(), g_i])
        flat_p.wait()
        # 6. dense-exact Adam: the looked-up rows of this rank's blocks, over the swept tables
        be.sweep_wait()
        if hasattr(be, "adam_finish_tables"):  # both tables' looked-up rows as one launch
            be.adam_finish_tables(self.hyper, [(self.users.weight, self.users.m, self.users.v, st_u, g_u),
                                               (self.items.weight, self.items.m, self.items.v, st_i, g_i)])
        else:
            be.adam_table_finish(self.users.weight, self.users.m, self.users.v, self.hyper, st_u, g_u)
            be.adam_table_finish(self.items.weight, self.items.m, self.items.v, self.hyper, st_i, g_i)
        be.adam_dense(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.hyper)
        self.comm_bytes = {"lookup_ids_alltoall": sent_ids, "lookup_rows_alltoall": sent_rows,
                           "rowgrad_alltoall": sent_rows,
                           "item_emb_allgather": (W - 1) * B * D * 4 if glob else 0,
                           "dI_reduce_scatter": (W - 1) * B * D * 4 if glob else 0,
                           "dense_grad_allreduce": int(2 * (W - 1) / W * self.flat_g.numel() * 4),
                           "scalars": 3 * 4 * (W - 1)}
        if self._scan is not None:
            self._scan.end()
        return loss

    def _weighted_loss(self, ce, labels, B, glob):
        """Value weights (ref :322,334-343) with the max / mean taken over the global batch -> (loss, dL/dce)."""
        denom = float(B * self.W)  # global negatives: mean over W*B rows; local: mean of W per-rank means
        if hasattr(self.be, "value_weights") and labels.dim() == 2 and labels.dtype == torch.float32 and _LOSS_KERNELS:
            # two launches around the two scalar exchanges instead of nine B-sized torch launches (csrc/inbatch_ce.hip)
            nuv, nmax = self.be.value_weights(labels, self.uvw)
            if glob:
                _timed_sync("scalars", all_reduce_, nmax, op=dist.ReduceOp.MAX)
            coef, loss = self.be.weighted_loss_global(nuv, nmax, ce, denom)
            _timed_sync("scalars", all_reduce_, loss)
            self.last_loss = loss
            return loss, coef
        nuv = torch.clamp(torch.sum(labels * self.uvw, dim=-1), min=0.000001)
        nmax = nuv.max()
        if glob:
            _timed_sync("scalars", all_reduce_, nmax, op=dist.ReduceOp.MAX)
        w = nuv / nmax
        coef = (w / denom).contiguous()
        loss = (ce * w).sum() / denom
        _timed_sync("scalars", all_reduce_, loss)
        self.last_loss = loss
        return loss, coef

    def _step_allgather(self, batch) -> torch.Tensor:
        """Round 1's fixed-size routing (A/B partner, `routing="allgather"`): all_gather of ids, every owner
        gathers W*B rows (zeros for foreign ids), reduce_scatter; row gradients all_gather'ed to every owner."""
        user_id, user_feat, hist_ids, item_id, item_feat, _pos, labels = batch
        be, W, D, B = self.be, self.W, self.cfg["D"], user_id.shape[0]
        # 1. embedding rows of the local batch, served by the owning ranks
        lk_u, u_emb_p = self._lookup(self.users, user_id)
        if self.hist:  # history rows first: the reference's lookup order on the item table
            lk_h, h_rows_p = self._lookup(self.items, hist_ids.reshape(-1))
        lk_i, i_emb_p = self._lookup(self.items, item_id)
        item_local = torch.cat([lk_h.local, lk_i.local]) if self.hist else lk_i.local
        # the tables' old rows have been read: plan, park the looked-up rows, and start the
        # zero-gradient sweep on the side stream -- it overlaps everything up to step 6
        if hasattr(be, "adam_begin_tables"):  # advance + both stashes as one launch
            st_u, st_i = be.adam_begin_tables(self.hyper, [(self.users.weight, self.users.m, self.users.v, lk_u.n_local, lk_u.local),
                                                           (self.items.weight, self.items.m, self.items.v, lk_i.n_local, item_local)])
        else:
            be.adam_advance(self.hyper)
            st_u = be.adam_table_begin(self.users.weight, self.users.m, self.users.v, lk_u.n_local, lk_u.local)
            st_i = be.adam_table_begin(self.items.weight, self.items.m, self.items.v, lk_i.n_local, item_local)
        sweep = [(self.users.weight, self.users.m, self.users.v, lk_u.n_local),
                 (self.items.weight, self.items.m, self.items.v, lk_i.n_local)]
        if not self._sweep_late:
            be.sweep_async(sweep, self.hyper, self._sweep_wgs)
        pu, pi = self._tower_params("user"), self._tower_params("item")
        summary, enc_saved = None, None
        u_emb = u_emb_p.wait()  # the item-side exchange is still in flight underneath the user tower
        if self.hist:
            enc_params = [self.params[k] for k in self.encoder_keys]
            summary3, enc_saved = be.encoder_fwd(h_rows_p.wait().view(B, -1, D), self.pe, self.heads, enc_params)
            summary = summary3.reshape(B, 2 * D)
        u_h, u_f, U = be.tower_fwd(u_emb, user_feat, pu, extra=summary)
        i_emb = i_emb_p.wait()
        i_h, i_f, I = be.tower_fwd(i_emb, item_feat, pi)
        # 2. logits against every rank's items
        glob = self.negatives == "global" and W > 1
        I_all = all_gather_rows(I) if glob else I
        off = self.rank * B if glob else 0
        if self._sweep_late:  # a short sweep hides under the logits kernels instead of the small tower GEMMs
            be.sweep_async(sweep, self.hyper, self._sweep_wgs)
        ce, lse = be.ce_fwd(U, I_all, off)
        # 3. value weights (ref :322,334-343) with the max / mean taken over the global batch
        loss, coef = self._weighted_loss(ce, labels, B, glob)
        # 4. backward through the loss
        dU, dI_all = be.ce_bwd(U, I_all, off, lse, coef)
        dI_p = reduce_scatter_rows_start(dI_all) if glob else _Pending(dI_all)  # travels under the user tower backward
        # 5. towers backward -> dense grads (flat buffer) + embedding-row grads
        # the row-gradient exchanges start as soon as their operand exists and run under what follows
        d_urows, d_summary = be.tower_bwd(dU, u_emb, u_h, u_f, user_feat, pu, self._tower_grads("user"), extra=summary)
        g_u_p = all_gather_rows_start(d_urows)  # aligned with lk_u.local
        g_h_p = None
        if self.hist:
            d_hrows = be.encoder_bwd(enc_saved, d_summary.view(B, 2, D), [self.grads[k] for k in self.encoder_keys])
            g_h_p = all_gather_rows_start(d_hrows)
        d_irows, _ = be.tower_bwd(dI_p.wait(), i_emb, i_h, i_f, item_feat, pi, self._tower_grads("item"))
        g_i_p = all_gather_rows_start(d_irows)
        if hasattr(be, "join_side"):
            be.join_side()
        flat_p = all_reduce_start_(self.flat_g)  # every dense gradient has been written by now
        g_u, g_i = g_u_p.wait(), g_i_p.wait()
        if self.hist:  # aligned with item_local = [history ids | item ids]
            g_i = torch.cat([g_h_p.wait(), g_i])
        flat_p.wait()
        # 6. dense-exact Adam: the looked-up rows of this rank's blocks, over the swept tables
        be.sweep_wait()
        if hasattr(be, "adam_finish_tables"):  # both tables' looked-up rows as one launch
            be.adam_finish_tables(self.hyper, [(self.users.weight, self.users.m, self.users.v, st_u, g_u),
                                               (self.items.weight, self.items.m, self.items.v, st_i, g_i)])
        else:
            be.adam_table_finish(self.users.weight, self.users.m, self.users.v, self.hyper, st_u, g_u)
            be.adam_table_finish(self.items.weight, self.items.m, self.items.v, self.hyper, st_i, g_i)
        be.adam_dense(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.hyper)
        n_rows_looked_up = B * (2 + (hist_ids.shape[1] if self.hist else 0))
        self.comm_bytes = {"lookup_ids_allgather": (W - 1) * n_rows_looked_up * 8,
                           "lookup_rows_reduce_scatter": (W - 1) * n_rows_looked_up * D * 4,
                           "rowgrad_allgather": (W - 1) * n_rows_looked_up * D * 4,
                           "item_emb_allgather": (W - 1) * B * D * 4 if glob else 0,
                           "dI_reduce_scatter": (W - 1) * B * D * 4 if glob else 0,
                           "dense_grad_allreduce": int(2 * (W - 1) / W * self.flat_g.numel() * 4),
                           "scalars": 3 * 4 * (W - 1)}
        return loss


# ----------------------------------------------------------------- sharded MIPS (BASELINE config 5)
class ShardedMIPS:
    """Brute-force MIPS over a corpus whose rows are split into W contiguous blocks
    (ref:src/baseline_mips_module.py:32-72 on one shard per GPU).  Every rank brings its own
    B queries; per call:
        all_gather queries                         [B, D] -> [W*B, D]
        local exact top-K of ALL queries on this rank's block (tt_mips_topk)
        all_to_all of the (score, global index) lists   [W, B, K] <-> [W, B, K]   (fixed size)
        exact merge of the W*K candidates per own query (tt_mips_merge)
    The global top-K is a subset of the union of the per-block top-Ks and every stage uses
    the (score desc, index asc) order, so the result equals the single-device answer."""

    def __init__(self, corpus_block: torch.Tensor, row_offset: int, backend=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedMIPS needs torch.distributed to be initialised")
        self.corpus, self.row_offset = corpus_block, int(row_offset)
        self.W = dist.get_world_size()
        self.be = backend if backend is not None else HipBackend(corpus_block.device)

    @staticmethod
    def block_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
        per = (n_rows + world - 1) // world
        lo = min(rank * per, n_rows)
        return lo, min(lo + per, n_rows)

    def search(self, query: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        W, B = self.W, query.shape[0]
        q_all = all_gather_rows(query) if W > 1 else query
        n_local = self.corpus.shape[0]
        k_loc = min(k, n_local)
        if k_loc > 0:
            idx, sc = self.be.mips_topk(q_all, self.corpus, k_loc)  # [W*B, k_loc], local row numbers
            idx = idx + self.row_offset
        else:  # this rank's block is empty (fewer corpus rows than ranks x rows per rank): "no candidate" only
            idx = torch.empty(q_all.shape[0], 0, dtype=torch.int64, device=q_all.device)
            sc = torch.empty(q_all.shape[0], 0, dtype=torch.float32, device=q_all.device)
        if k_loc < k:  # a block smaller than K: pad with "no candidate"
            pad = k - k_loc
            idx = torch.cat([idx, idx.new_full((idx.shape[0], pad), -1)], dim=1)
            sc = torch.cat([sc, sc.new_zeros((sc.shape[0], pad))], dim=1)
        if W > 1:
            ridx, rsc = all_to_all_rows(idx), all_to_all_rows(sc)  # chunk r of the send = rank r's queries
            # received layout [W (source shard), B, k] -> per own query the W*k candidates
            idx = ridx.view(W, B, k).permute(1, 0, 2).reshape(B, W * k)
            sc = rsc.view(W, B, k).permute(1, 0, 2).reshape(B, W * k)
        return self.be.mips_merge(sc, idx, k)
