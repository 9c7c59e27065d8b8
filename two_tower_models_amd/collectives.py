"""Collectives of the row-sharded step (one process per GPU; backend "nccl" IS RCCL on ROCm).

The reference has no communication of any kind: everything here is new design.  Two
transports behind the same helpers: torch.distributed's process group (default), or the C ABI's own tt_comm_*
(comm.NativeComm, `use_native_transport`) -- RCCL bound by libtt_hotpath.so itself.  `*_start` helpers return a
`_Pending` whose `.wait()` makes the CURRENT stream wait for the result, so kernels queued in between overlap the
exchange; every exchange carries a tag for the timing summary (bench.py: `comm.ms_per_step`) and for the watchdog's
report when one does not complete.  gloo (tests on a 1-GPU or CPU-only box) moves host memory only: device tensors
are staged through the host there, never under RCCL.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _native as _N

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
    """gloo moves host memory only.  Device tensors under gloo (several ranks sharing ONE GPU: how
    tests/test_gpu_parallel.py runs the product kernels at world size > 1 on a 1-GPU box) are
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


# ----------------------------------------------------------------- first contact with N GPUs: who hung, on what
# Every exchange is noted (sequence number, tag, bytes) in a short ring.  RCCL collectives are asynchronous: a rank that
# never arrives shows up as a host wait that does not return (`.item()`, synchronize) with nothing on the screen.  The
# Watchdog turns that into an error message: the training / bench loop marks each step with an event; a daemon thread
# polls the oldest unfinished mark and, when it is older than `seconds`, prints the exchanges issued since the last step
# that DID complete (tag, bytes, in order -- the first one is where the group is stuck) and ends the process.
_EXCHANGES: list = []
_EXCHANGE_SEQ = [0]


def note_exchange(tag: str, x) -> None:
    _EXCHANGE_SEQ[0] += 1
    n = x.numel() * x.element_size() if isinstance(x, torch.Tensor) else 0
    _EXCHANGES.append((_EXCHANGE_SEQ[0], tag, n))
    if len(_EXCHANGES) > 256:
        del _EXCHANGES[:128]


def exchanges_since(seq: int):
    return [e for e in _EXCHANGES if e[0] > seq]


class Watchdog:
    """`wd = Watchdog(30); ... wd.mark() once per step ...; wd.close()`.  `on_timeout(report)` replaces the default
    (print to stderr, os._exit(124)) -- tests use it."""

    def __init__(self, seconds: float = 30.0, on_timeout=None, poll: float = 0.5):
        self.seconds, self.on_timeout, self._poll = float(seconds), on_timeout, poll
        self._marks: list = []  # (event, host time, exchange sequence number at the mark, step)
        self._done_seq, self._done_step, self._step = 0, 0, 0
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self.fired = False
        self._thread = threading.Thread(target=self._run, name="tt-comm-watchdog", daemon=True)
        self._thread.start()

    def mark(self, event=None) -> None:
        """Call after a step has been enqueued (any stream-ordered point works)."""
        if event is None:
            event = torch.cuda.Event()
            event.record()
        self._step += 1
        with self._lock:
            self._marks.append((event, time.monotonic(), _EXCHANGE_SEQ[0], self._step))

    def report(self) -> str:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        pend = exchanges_since(self._done_seq)
        lines = [f"[tt watchdog] rank {rank}/{world}: no step has completed for {self.seconds:.0f} s "
                 f"(last completed step {self._done_step}, {len(self._marks)} enqueued behind it).",
                 "  exchanges issued since the last completed step, oldest first (the group is most likely stuck in the first):"]
        lines += [f"    #{seq} {tag} ({n} bytes)" for seq, tag, n in pend[:24]] or ["    (none: the hang is not in a collective)"]
        return "\n".join(lines)

    def _run(self) -> None:
        while not self._stop.wait(self._poll):
            with self._lock:
                while self._marks and self._marks[0][0].query():
                    _, _, self._done_seq, self._done_step = self._marks.pop(0)
                stuck = bool(self._marks) and time.monotonic() - self._marks[0][1] > self.seconds
            if stuck:
                self.fired = True
                text = self.report()
                if self.on_timeout is not None:
                    self.on_timeout(text)
                    return
                sys.stderr.write(text + "\n")
                sys.stderr.flush()
                os._exit(124)

    def close(self) -> None:
        self._stop.set()


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
        if tag is not None:
            note_exchange(tag, out)
            if _N.trace is not None:
                _N.trace.append("issue:" + tag)

    def wait(self) -> torch.Tensor:
        if _N.trace is not None and self.tag is not None:
            _N.trace.append("wait:" + self.tag)
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
    note_exchange(tag, args[0] if args else None)
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

