"""DenseExactAdam: torch.optim.Adam semantics for the two-tower trainer, on HIP.

``optim.Adam(model.parameters())`` in the reference loop (ref:train/train.py:179,
:123-125) updates EVERY row of both embedding tables every step, because
``nn.Embedding`` produces a dense gradient.  This optimiser is value-equivalent
(same update rule, same bias correction, every row stepped) but never builds the
dense gradient: the embedding backward leaves its row gradients with the table
(``weight._tt_rowgrads``), and the table step is

    plan (stable sort of the looked-up ids)  ->  Adam on the looked-up rows
    ->  zero-gradient Adam sweep over the whole table (the HBM-bound part)
    ->  write the looked-up rows back,

one multi-tensor launch for all dense parameters, and one 1-thread launch that
advances the step count / bias corrections in device memory.

Overlapped schedule (``overlap_sweep=True``, default).  The sweep does not depend on the
step's gradients, only on the forward's lookups having read the old rows.  The lookups
register their ids with the table at forward time, so when the caller follows the
reference loop order

    loss = model.train_forward(...); opt.zero_grad(); loss.backward(); opt.step()

``zero_grad()`` -- which sits between forward and backward -- already knows the rows:
it plans, parks the old p/m/v of the looked-up rows in a side buffer and launches the sweep
on a second HIP stream, where it runs concurrently with the backward pass (HBM-bound next
to MFMA/latency-bound work).  ``step()`` then updates the looked-up rows from the side
buffer and writes them over the swept table.  Results are bit-identical to the serial
schedule.  Any other call order (zero_grad before forward, no zero_grad at all) simply
takes the serial schedule inside ``step()``.  The one thing the overlapped schedule
assumes is that a ``zero_grad()`` issued after a forward IS followed by that forward's
``backward()`` and ``step()``; pass ``overlap_sweep=False`` if that does not hold.

``overlap_sweep="forward"`` goes one step further: the models' ``train_forward`` announces the
ids it is about to look up (``begin_step``) BEFORE touching the tables.  The optimiser plans,
parks the old rows straight from the id lists (no sort needed yet), starts the sweep at once and
hands the lookups a view of the parked rows (``ops.ActiveStash``), so the id sort, forward AND
backward all run under the sweep.  It assumes every
``train_forward`` executed with autograd enabled is followed by ``backward()`` and ``step()``
(exactly the reference loop); results are again bit-identical to the serial schedule.

Deferred schedule (``lazy=True``; SURVEY 8f-3, reported separately from the dense figure).  The
zero-gradient steps of an untouched row are a recurrence on the row alone, so instead of sweeping
the table every step the optimiser REPLAYS them, in registers and with the same fp32 operations
in the same order, when a row is next looked up, updated or flushed.  Tables end up bit-identical
to the dense schedule (``tests/test_gpu_models.py::test_lazy_adam_is_bit_identical_to_dense``);
the step no longer streams 24 B per table element.  Lookups through this package's modules catch
their rows up automatically; anything else that reads a table as a whole (``model.state_dict()``,
exporting a corpus) must call ``optimizer.flush()`` first (``optimizer.state_dict()`` does).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from . import _native as N
from . import ops


_RELEASE_AT_LOGITS = True  # a sweep held for the backward logits kernel is released BY that kernel's Function (A/B switch)
_SPLIT_MIN_IDS = 65536  # looked-up rows per step from which the moments leave the main stream's begin launch (tests lower it)
_MARK_ROWS = True  # ... and are not parked at all: the sweep steps over the step's rows (tools/ab_c3.py flips it)
_HOLD_SWEEP = True  # (tools/ab_c3.py flips it: the held-back sweep start against the immediate one, same process, same box)


class _TableStep:
    """State of one table between the overlapped begin (in zero_grad) and finish (in step)."""

    __slots__ = ("plan", "side", "announced", "marked", "n_rows")

    def __init__(self, plan, side, announced=False, marked=False, n_rows=0):
        self.plan, self.side, self.announced = plan, side, announced
        self.marked = marked  # the looked-up rows are marked for the sweep to step over, not parked (_begin_overlapped): no side buffer
        self.n_rows = n_rows  # rows of the table this process owns


class _LazyRows:
    """Attached to a table as ``weight._tt_lazy``: lets a lookup bring its rows up to date."""

    __slots__ = ("opt", "param")

    def __init__(self, opt: "DenseExactAdam", param: torch.nn.Parameter):
        self.opt, self.param = weakref.ref(opt), param

    def catch_up(self, ids: torch.Tensor) -> None:
        opt = self.opt()
        if opt is not None:
            opt._catch_up(self.param, ids)


class DenseExactAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, overlap_sweep=True, lazy: bool = False) -> None:
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError("DenseExactAdam supports a single parameter group")
        self._params: List[torch.nn.Parameter] = [p for p in self.param_groups[0]["params"]]
        self._tables = [p for p in self._params if getattr(p, "_tt_is_table", False)]
        self._dense = [p for p in self._params if not getattr(p, "_tt_is_table", False)]
        for p in self._tables:
            p._tt_rowgrads = []  # switches the embedding backward to row form
            p._tt_lookups = []   # the forward's lookups register their ids here
            p._tt_active = None  # ops.ActiveStash while a "forward"-mode step is under way
            p._tt_optimizer = weakref.ref(self)
        if overlap_sweep not in (True, False, "forward"):
            raise ValueError('overlap_sweep must be True, False or "forward"')
        # Row-sharded tables (parallel.py): this rank's Parameter is its row block, the state below its moments.  The
        # looked-up rows are known to their OWNER only through the lookup exchange, which train_forward starts -- so the
        # forward-announced schedule is the one that applies; the replicated parameters' gradients are all-reduced (SUM,
        # one flat buffer) in step().
        self._sharded = [p for p in self._tables if getattr(p, "_tt_shard", None) is not None]
        if self._sharded:
            if len(self._sharded) != len(self._tables):
                raise ValueError("either every embedding table of the model is row-sharded or none is")
            if lazy:
                raise ValueError("the deferred (lazy) schedule is a single-GPU schedule; row-sharded tables use the dense-exact sweep")
            overlap_sweep = "forward"
        self._flat_g: Optional[torch.Tensor] = None
        self.lazy = bool(lazy)
        self.overlap_sweep = False if self.lazy else overlap_sweep  # no sweep to overlap
        self._last_step: Dict[int, torch.Tensor] = {}
        self._tab: Optional[torch.Tensor] = None
        self._tab_steps = 0
        self._host_steps = 0  # steps taken, counted on the host (no device sync)
        self._resume_step = 0  # step count adopted from a loaded checkpoint
        self._prefetch_done: Optional[torch.cuda.Event] = None
        self._prefetch_keep = None
        if self.lazy:
            for p in self._tables:
                p._tt_lazy = _LazyRows(self, p)
        self._hyper = None
        self._ready = False
        self._side_stream: Optional[torch.cuda.Stream] = None
        self.capture_overlap = False  # set by graphs.GraphedTrainStep (multi-stream capture of the overlapped schedule)
        self._sweep_wgs = 0  # 0 = library default (3 workgroups per CU); lowered by the throttle controller
        self._tune = None  # events of the step in flight: [begin, sweep start, sweep end, end, level, step number]
        self._tune_done: List[list] = []  # finished steps whose events may still be pending on the GPU
        self._tune_state = None  # the sweep-level scan (see _tune_sweep)
        self._plan_stream: Optional[torch.cuda.Stream] = None
        self._plan_done: Optional[torch.cuda.Event] = None
        self._plans_pending = False
        self._plan_ready: Optional[torch.cuda.Event] = None
        self._begun: Optional[Dict[torch.nn.Parameter, _TableStep]] = None
        self._sweep_done: Optional[torch.cuda.Event] = None
        self._side_bufs: Dict[int, torch.Tensor] = {}
        self._marks: Dict[int, torch.Tensor] = {}  # table -> one bit per row: the rows this step looks up (marked schedule)
        self._sweep_pending = None  # a sweep held back for a later point of the step (release_sweep)
        self._sweep_pending_marked = False
        self._sweep_events = None  # keep_sweep_events(): (start, end) event pairs of the table sweep launches
        self._catchup_last: Dict[int, tuple] = {}  # deferred schedule: table -> (event, stream) of its latest catch-up this step

    # state is created lazily, on the parameters' device
    def _init_state(self) -> None:
        dev = N.require_device(*self._params)
        g = self.param_groups[0]
        self._hyper = torch.tensor([g["lr"], g["betas"][0], g["betas"][1], g["eps"], 0.0, 0.0, 0.0, 0.0],
                                   dtype=torch.float64, device=dev)
        arena = bool(self._tables) and all(p.is_cuda for p in self._tables) and os.environ.get("TT_ADAM_NO_ARENA") is None
        in_arena = {id(p) for p in self._tables} if arena else set()
        for p in self._params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("DenseExactAdam needs contiguous fp32 parameters")
            st = self.state[p]
            for key in ("exp_avg", "exp_avg_sq"):  # a loaded checkpoint already supplied them
                if key not in st and id(p) in in_arena:
                    continue  # born as zeros inside the arena below
                if key not in st or st[key].shape != p.shape or st[key].device != p.device:
                    st[key] = torch.zeros_like(p) if key not in st else st[key].to(p.device, torch.float32).contiguous()
        if arena:
            self._home_tables_in_one_arena()
        self._side_stream = N.low_priority_stream(dev)
        start = int(self._resume_step)
        self._hyper[4] = float(start)  # [5], [6] are recomputed from the step by every advance
        self._host_steps = start
        if self.lazy:
            # a checkpoint holds flushed tables: every row is current for the step it was taken at
            for p in self._tables:
                self._last_step[id(p)] = torch.full((p.shape[0],), start, dtype=torch.int32, device=dev)
            self._tab_steps = max(1 << 16, 2 * (start + 2))
            self._tab = torch.zeros(2 * self._tab_steps, dtype=torch.float32, device=dev)
        self._ready = True

    def _home_tables_in_one_arena(self) -> None:
        """Every table's p, m, v in ONE device allocation, back to back on 2 MiB boundaries (the Parameter keeps its identity:
        `p.data` becomes a view of the arena) -- and the allocation itself CHOSEN BY MEASUREMENT.  What the sweep streams at is
        a property of the allocation it runs over, decided when the driver backs it with physical pages, and it differs by
        7-8 % between allocations of one process on one box (profiles/r06_sweep_placement.txt: 40 random placements of the six
        arrays inside one allocation 6.56-6.62 TB/s, six separate allocations of the same process 6.0-6.15; two bench.py
        processes on one box 5.39 and 5.85 ms per step -- the "slow mode" of the driver's round-5 line; the streaming copy
        beside it does not move).  So up to TT_ADAM_ARENA_TRIES (default 6: three of four candidates of one process have been seen slow) candidate arenas are held at once, the sweep
        kernel itself is timed over each (three launches on uninitialised memory: values do not matter, it is overwritten
        below), the fastest one is kept and the others go back to the driver.  Costs ~0.1 s and one transient copy of the
        tables at the first step; skipped for candidates that do not fit next to each other (C4: one 154 GB arena).
        A caller that saves `model.state_dict()` on its own afterwards stores the arena's bytes (tables + moments);
        `{k: v.clone() for k, v in model.state_dict().items()}` stores the tables only."""
        A = 2 << 20
        sizes = [(p.numel() * 4 + A - 1) // A * A for p in self._tables]
        nbytes = 3 * sum(sizes) + A
        dev = self._tables[0].device

        def views(arena):
            off, out = (-arena.data_ptr()) % A, []
            for p, size in zip(self._tables, sizes):
                row = []
                for _ in range(3):
                    row.append(arena[off:off + p.numel() * 4].view(torch.float32).view(p.shape))
                    off += size
                out.append(row)
            return out

        tries = max(1, int(os.environ.get("TT_ADAM_ARENA_TRIES", "6")))
        free, _total = torch.cuda.mem_get_info(dev)
        tries = max(1, min(tries, int((free * 0.8) // nbytes)))
        cands = []
        try:
            for _ in range(tries):
                cands.append(torch.empty(nbytes, dtype=torch.uint8, device=dev))
        except torch.OutOfMemoryError:
            pass
        if not cands:  # no room even for the transient second copy of the tables: they stay where they are
            for p in self._tables:
                for key in ("exp_avg", "exp_avg_sq"):
                    self.state[p].setdefault(key, torch.zeros_like(p))
            return
        rates = []
        if len(cands) > 1:
            lib = N.load()
            lib.tt_profile_pause(1)  # these launches are this method's own measurement, not a caller's (tt_profile_*)
            for arena in cands:
                vs = views(arena)
                descs = (N.AdamTensor * len(vs))()
                for i, (w, m, v) in enumerate(vs):
                    descs[i].p, descs[i].g, descs[i].m, descs[i].v, descs[i].n = w.data_ptr(), None, m.data_ptr(), v.data_ptr(), w.numel()
                evs = []
                for _ in range(4):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    N.check(lib.tt_adam_tables_sweep(descs, len(vs), self._hyper.data_ptr(), 0, N.stream()), "tt_adam_tables_sweep")
                    e1.record()
                    evs.append((e0, e1))
                evs[-1][1].synchronize()
                ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs[1:])  # (the first launch touches the pages)
                rates.append(24.0 * sum(p.numel() for p in self._tables) / (ms[0] * 1e-3) / 1e9)
            lib.tt_profile_pause(0)
            best = max(range(len(cands)), key=lambda i: rates[i])
        else:
            best = 0
        arena = cands[best]
        del cands
        if rates:
            torch.cuda.empty_cache()  # the losing candidates go back to the DRIVER, not into torch's block cache
        self.arena_note = {"candidates_GBps": [round(r, 1) for r in rates], "kept": best, "bytes": nbytes}
        with torch.no_grad():
            for p, (home, m, v) in zip(self._tables, views(arena)):
                st = self.state[p]
                home.copy_(p.data)
                p.data = home
                for key, t in (("exp_avg", m), ("exp_avg_sq", v)):
                    if key in st:
                        t.copy_(st[key])  # what a loaded checkpoint supplied
                    else:
                        t.zero_()
                    st[key] = t
        self._arena = arena

    # ------------------------------------------------------------------ deferred schedule
    def _catch_up(self, p: torch.nn.Parameter, ids: torch.Tensor) -> None:
        """Rows `ids` of table `p` -> current step (called by the lookups before they read)."""
        if not self._ready or self._host_steps == 0 or not p.is_cuda or ids.numel() == 0:
            return  # nothing has been deferred yet (fresh, or just loaded from a flushed checkpoint)
        cur = torch.cuda.current_stream()
        if self._prefetch_done is not None:  # never replay a row on two streams at once (step() drops the event: a forward
            cur.wait_event(self._prefetch_done)  # whose towers run on two streams waits on both)
        if ops._aux_forks[0]:
            # ... nor by two towers that look up the SAME table from different streams (ops.AuxFork: the history model's
            # user tower and its item tower both read the item table): per table, a catch-up waits for the previous one
            prev = self._catchup_last.get(id(p))
            if prev is not None and prev[1] != cur:
                cur.wait_event(prev[0])
        ids = ids.reshape(-1)
        if ids.dtype != torch.int64 or not ids.is_contiguous():
            ids = ids.to(torch.int64).contiguous()
        st = self.state[p]
        N.check(N.load().tt_adam_rows_catchup(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                              p.shape[0], p.shape[1], ids.data_ptr(), ids.numel(),
                                              self._last_step[id(p)].data_ptr(), self._hyper.data_ptr(),
                                              self._tab.data_ptr(), self._tab_steps, N.stream()),
                "tt_adam_rows_catchup")
        if ops._aux_forks[0]:
            ev = torch.cuda.Event()
            ev.record(cur)
            self._catchup_last[id(p)] = (ev, cur)

    @torch.no_grad()
    def prefetch_rows(self, lookups: Dict[torch.nn.Parameter, Sequence[torch.Tensor]]) -> None:
        """Deferred schedule: replay the idle steps of the rows a LATER batch will look up, on the
        low-priority side stream, underneath the current step's forward / backward (the replay is
        VALU work, the step is MFMA work).  Call it after the current step's ``train_forward`` --
        rows the current step reads are then already current and are left alone -- with
        ``model._lookup_plan(user_id, user_history, item_id)`` of the next batch.  ``step()`` waits
        for it.  Purely a scheduling hint: results are bit-identical with or without it."""
        if not self.lazy or not self._ready or self._host_steps == 0:
            return
        mine = {id(p): p for p in self._tables}
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)  # the current step's own catch-ups and gathers are enqueued before this point
        self._side_stream.wait_event(ready)
        lib = N.load()
        keep = []
        for p, blocks in lookups.items():
            if id(p) not in mine or not p.is_cuda:
                continue
            st = self.state[p]
            for ids in blocks:
                ids = ids.reshape(-1)
                if ids.dtype != torch.int64 or not ids.is_contiguous():
                    ids = ids.to(torch.int64).contiguous()
                if ids.numel() == 0:
                    continue
                keep.append(ids)
                N.check(lib.tt_adam_rows_catchup(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                 p.shape[0], p.shape[1], ids.data_ptr(), ids.numel(),
                                                 self._last_step[id(p)].data_ptr(), self._hyper.data_ptr(),
                                                 self._tab.data_ptr(), self._tab_steps,
                                                 self._side_stream.cuda_stream), "tt_adam_rows_catchup")
        self._prefetch_done = torch.cuda.Event()
        self._prefetch_done.record(self._side_stream)
        self._prefetch_keep = keep  # the id tensors must outlive the side-stream kernels

    @torch.no_grad()
    def flush(self) -> None:
        """Deferred schedule: bring EVERY table row up to the current step.  Needed before the
        tables are read other than through this package's lookups (checkpoint, corpus export)."""
        if not self.lazy or not self._ready or self._host_steps == 0:
            return
        if self._prefetch_done is not None:
            torch.cuda.current_stream().wait_event(self._prefetch_done)
            self._prefetch_done, self._prefetch_keep = None, None
        lib = N.load()
        for p in self._tables:
            st = self.state[p]
            N.check(lib.tt_adam_table_flush(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                            p.shape[0], p.shape[1], self._last_step[id(p)].data_ptr(),
                                            self._hyper.data_ptr(), self._tab.data_ptr(), self._tab_steps,
                                            N.stream()), "tt_adam_table_flush")

    def state_dict(self):
        """torch.optim.Adam's layout (per parameter: step, exp_avg, exp_avg_sq), tables flushed."""
        self.flush()
        sd = super().state_dict()
        step = torch.tensor(float(self.step_count))
        for st in sd["state"].values():
            st["step"] = step.clone()
        return sd

    def load_state_dict(self, state_dict) -> None:
        """Accepts this class's or torch.optim.Adam's state (same keys).  Takes effect at the next
        step: the moments are adopted as they are, the step count restarts from the stored one."""
        super().load_state_dict(state_dict)
        steps = {int(float(st["step"])) for st in self.state.values() if "step" in st}
        if len(steps) > 1:
            raise ValueError("DenseExactAdam needs one common step count for all parameters")
        self._resume_step = steps.pop() if steps else 0
        self._begun = None
        self._ready = False  # device-side state is rebuilt around the loaded moments

    def _advance_lazy(self) -> None:
        if self._host_steps + 2 >= self._tab_steps:  # grow the per-step constant table (x2)
            bigger = torch.zeros(4 * self._tab_steps, dtype=torch.float32, device=self._tab.device)
            bigger[: 2 * self._tab_steps].copy_(self._tab)
            self._tab, self._tab_steps = bigger, 2 * self._tab_steps
        N.check(N.load().tt_adam_advance_tab(self._hyper.data_ptr(), self._tab.data_ptr(), self._tab_steps,
                                             N.stream()), "tt_adam_advance_tab")
        self._host_steps += 1

    @property
    def step_count(self) -> int:
        if not self._ready:  # fresh, or a checkpoint was loaded and no step has run since
            return int(self._resume_step)
        return int(self._hyper[4].item())

    def _side(self, p: torch.Tensor, nbytes: int) -> torch.Tensor:
        buf = self._side_bufs.get(id(p))
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=p.device)
            self._side_bufs[id(p)] = buf
        return buf

    # ------------------------------------------------------------------ overlapped begin
    def _begin_overlapped(self, announced: Optional[Dict[int, Sequence[torch.Tensor]]] = None, hold_sweep: bool = False) -> None:
        """Plan + park the old rows on the main stream, then launch the sweep on the side stream.
        `announced` (forward mode): the id blocks each table WILL be looked up with, in lookup
        order; otherwise the blocks the forward already registered."""
        lib = N.load()
        if not self._ready:
            self._init_state()
        hyper = self._hyper.data_ptr()
        capturing = torch.cuda.is_current_stream_capturing()  # whole-step hipGraph: no timing events, no controller
        ev_begin = None
        if not capturing:
            self._tune_sweep()
            ev_begin = torch.cuda.Event(enable_timing=True)
            ev_begin.record()
            if self._tune_done and self._tune_done[-1][5] == self._host_steps and len(self._tune_done[-1]) == 6:
                self._tune_done[-1].append(ev_begin)  # the previous step's record: its full period ends where this step begins
        if announced is None:
            N.check(lib.tt_adam_advance(hyper, N.stream()), "tt_adam_advance")
        self._host_steps += 1
        begun: Dict[torch.nn.Parameter, _TableStep] = {}
        stash_jobs = []  # forward mode: the step-count advance and every table's stash go out as ONE launch
        todo = []  # (table, id blocks, rows it owns)
        for p in self._tables:
            blocks = announced.get(id(p)) if announced is not None else p._tt_lookups
            if not blocks:
                continue
            shard = getattr(p, "_tt_shard", None)
            # a row block: `blocks` are the local ids the lookup exchange delivered, with the sentinel n_local for padding
            # slots -- the plan sorts it last (n_rows + 1 "rows") and the Adam kernels skip its run
            n_rows = p.shape[0] if shard is None else shard.n_local
            if n_rows > 0:  # (fewer rows than ranks: nothing to park, sweep or finish on this rank)
                todo.append((p, blocks, n_rows))
        # Many looked-up rows (history model: 217 K, 0.67 GB of p / m / v to park): only the p plane is needed before the
        # forward can start -- the moments are parked on the sweep's stream, in front of the sweep (C3: 0.15 ms at the head
        # of the step become 0.05).  Small lookups keep the single launch.
        n_stashed = sum(sum(b.numel() for b in blocks) for _, blocks, _ in todo)
        split_planes = announced is not None and not capturing and 0 < len(todo) <= 4 and n_stashed >= _SPLIT_MIN_IDS
        # ... and nothing is parked at all where the sweep can step over the looked-up rows instead (a bitmap of the step's
        # rows, tt_adam_mark_rows): no gather in front of the sweep, no sweep traffic for rows the finish overwrites; the
        # lookups read the table, whose marked rows keep their old values until the finish, and the finish reads p, m, v there
        marked = (split_planes and _MARK_ROWS and all(lib.tt_adam_marked_supported(p.shape[1]) for p, _, _ in todo)
                  and all((p.data_ptr() | self.state[p]["exp_avg"].data_ptr() | self.state[p]["exp_avg_sq"].data_ptr()) % 16 == 0
                          for p, _, _ in todo))
        for p, blocks, n_rows in todo:
            dim = p.shape[1]
            shard = getattr(p, "_tt_shard", None)
            st = self.state[p]
            # forward mode: the ids alone are enough to park the rows (slot = occurrence), so the
            # sort is deferred until the sweep is running
            plan = ops.RowPlan(blocks, n_rows + (1 if shard is not None else 0), slot=f"plan{id(p)}", defer=announced is not None)
            side = None if marked else self._side(p, lib.tt_adam_table_workspace_bytes(plan.n, dim))
            if marked:
                if shard is None:  # (a sharded table's lookups were served from the table before this point: parallel.begin_lookups)
                    p._tt_active = ops.ActiveMarks(plan.block_sizes)
            elif announced is not None:
                stash_jobs.append((p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n_rows, dim,
                                   plan.ids.data_ptr(), plan.n, side.data_ptr(), side.numel()))
                if shard is None:
                    p_plane = side[: plan.n * dim * 4].view(torch.float32).view(plan.n, dim)
                    p._tt_active = ops.ActiveStash(p_plane, plan.block_sizes)
            else:
                N.check(lib.tt_adam_table_stash(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                n_rows, dim, plan.n, plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                                plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), side.data_ptr(),
                                                side.numel(), N.stream()), "tt_adam_table_stash")
            begun[p] = _TableStep(plan, side, announced is not None, marked, n_rows)
        if marked:
            N.check(lib.tt_adam_advance(hyper, N.stream()), "tt_adam_advance")
        elif announced is not None:
            jobs = (N.AdamStashJob * max(len(stash_jobs), 1))()
            for i, j in enumerate(stash_jobs):
                (jobs[i].W, jobs[i].M, jobs[i].V, jobs[i].n_rows, jobs[i].dim, jobs[i].ids, jobs[i].n_ids, jobs[i].side,
                 jobs[i].side_bytes) = j
            if split_planes:
                N.check(lib.tt_adam_begin_ids_planes(hyper, None, 0, jobs, len(stash_jobs), 1, N.stream()), "tt_adam_begin_ids_planes")
            else:
                N.check(lib.tt_adam_begin_ids(hyper, None, 0, jobs, len(stash_jobs), N.stream()), "tt_adam_begin_ids")
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)  # lookups (or none yet, forward mode) + stashes are complete here

        def launch_sweep(held_back=False, after=None, ready=ready, begun=begun, split_planes=split_planes, capturing=capturing,
                         marked=marked):
            self._side_stream.wait_event(ready)
            if held_back:  # ... and after what was queued since (the big gather this was held back for) / after `after`
                later = after
                if later is None:
                    later = torch.cuda.Event()
                    later.record(torch.cuda.current_stream())
                self._side_stream.wait_event(later)
            mark_ptrs = None
            if marked:
                mark_ptrs = (C.c_void_p * len(begun))()
                for i, (p, ts) in enumerate(begun.items()):
                    n_local = ts.n_rows
                    words = lib.tt_adam_marks_words(n_local)
                    bm = self._marks.get(id(p))
                    if bm is None or bm.numel() < words:
                        bm = self._marks[id(p)] = torch.empty(words, dtype=torch.int32, device=p.device)
                    N.check(lib.tt_adam_mark_rows(ts.plan.ids.data_ptr(), ts.plan.n, n_local, bm.data_ptr(), bm.numel(),
                                                  self._side_stream.cuda_stream), "tt_adam_mark_rows")
                    mark_ptrs[i] = bm.data_ptr()
            elif split_planes:
                N.check(lib.tt_adam_begin_ids_planes(hyper, None, 0, jobs, len(stash_jobs), 6, self._side_stream.cuda_stream),
                        "tt_adam_begin_ids_planes")
            ev_s0 = None
            if not capturing:
                ev_s0 = torch.cuda.Event(enable_timing=True)
                ev_s0.record(self._side_stream)
            if begun:  # ONE launch for all tables: no gap and a single tail between the user and the item table
                descs = (N.AdamTensor * len(begun))()
                for i, p in enumerate(begun):
                    st = self.state[p]
                    descs[i].p, descs[i].g = p.data_ptr(), None
                    shard = getattr(p, "_tt_shard", None)
                    descs[i].m, descs[i].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    descs[i].n = p.numel() if shard is None else shard.n_local * p.shape[1]
                if mark_ptrs is not None:
                    dims = (C.c_int64 * len(begun))(*[p.shape[1] for p in begun])
                    N.check(lib.tt_adam_tables_sweep_marked(descs, dims, mark_ptrs, len(begun), hyper, self._sweep_wgs,
                                                            self._side_stream.cuda_stream), "tt_adam_tables_sweep_marked")
                else:
                    N.check(lib.tt_adam_tables_sweep(descs, len(begun), hyper, self._sweep_wgs, self._side_stream.cuda_stream),
                            "tt_adam_tables_sweep")
            self._sweep_done = torch.cuda.Event(enable_timing=not capturing)
            self._sweep_done.record(self._side_stream)
            if self._tune is not None:
                self._tune[1], self._tune[2] = ev_s0, self._sweep_done
            if self._sweep_events is not None and ev_s0 is not None and len(self._sweep_events) < 65536:
                self._sweep_events.append((ev_s0, self._sweep_done))

        # A step that gathers MANY rows right after this point (history model: B*H = 205 K random 512-B rows) lets that
        # gather run BEFORE the sweep starts saturating HBM: next to the sweep it took 281 us instead of 105 (round 4
        # profile), on the critical path.  The gather's Function calls release_sweep() once it is queued; zero_grad() /
        # step() do if nobody did.  The sweep is a third of such a step: starting it 0.1 ms later costs nothing.
        self._tune = None if capturing else [ev_begin, None, None, None, self._sweep_wgs, self._host_steps]
        self._sweep_done = None
        n_announced = sum(ts.plan.n for ts in begun.values()) if announced is not None else 0
        # ... and a caller whose logits kernels ARE the step (row-sharded tables from W = 4: thin row blocks, W*B negatives
        # per user) asks for the sweep to start with the BACKWARD logits kernel (zero_grad() releases it): the forward
        # logits kernel then runs at 0.81 instead of 0.73 of the matrix pipe, the backward one -- which streams the kept
        # logits from HBM anyway -- loses less than that (round 4: 4.10 vs 4.26 ms per emulated W = 8 step)
        if _HOLD_SWEEP and announced is not None and not capturing and ((n_announced >= 65536 and not self._sharded) or hold_sweep):
            self._sweep_pending = launch_sweep
            self._sweep_pending_marked = marked
            # hold_sweep: until the BACKWARD logits kernel is in the main stream's queue (ops.InBatchSoftmaxCE.backward
            # releases it, ordered behind an event recorded in front of that kernel): both become runnable at the same
            # moment and the kernel, already queued on the normal-priority stream, takes its workgroup slots first.  Released
            # by zero_grad() -- i.e. queued BEFORE the backward pass was -- the sweep's 256 workgroups were on the CUs first
            # whenever the host ran just ahead of the GPU, and the logits kernel ran at half occupancy next to them: emulated
            # W = 8 step 4.7 instead of 4.2 ms.  (bench.py's profiling events happened to delay the sweep the same way, which
            # is why the bench line never showed it: round 5.)
            self._hold_for_logits = bool(hold_sweep) and _RELEASE_AT_LOGITS
            if self._hold_for_logits:
                ops.held_sweeps.add(self)
        else:
            self._sweep_pending = None
            launch_sweep()
        # forward mode: the stable sort of the ids is needed only by finish (in step()).  It is NOT enqueued here: its
        # ~30 short launches would sit in front of the forward's kernels on the HOST (0.25 ms of enqueue time per step
        # -- at C2 the main stream idled that long before its first forward kernel).  zero_grad() -- after the forward
        # has been enqueued -- launches it on a third stream, next to the backward kernels; step() does if nobody did.
        self._plan_done = None
        self._plans_pending = announced is not None and bool(begun)
        self._plan_ready = ready if self._plans_pending else None
        self._begun = begun

    def keep_sweep_events(self, on: bool = True) -> None:
        """Keep the event pair every table sweep launch is bracketed by anyway (the sweep-width controller's, on the sweep's
        own stream) so that a caller can report the sweep's launch duration WITHOUT adding events of its own -- an extra
        pair in front of a kernel that is the step's critical path is not free (C2: 1.17 vs 1.09 ms per step)."""
        self._sweep_events = [] if on else None

    def sweep_launch_ms(self):
        """-> (total ms, launches) over the kept events; waits for them."""
        total, n = 0.0, 0
        for a, b in self._sweep_events or ():
            b.synchronize()
            total += a.elapsed_time(b)
            n += 1
        return total, n

    def held_sweep_is_marked(self) -> bool:
        """A sweep is held back AND it will step over this step's looked-up rows (so the lookups need not be queued first)."""
        return getattr(self, "_sweep_pending", None) is not None and bool(getattr(self, "_sweep_pending_marked", False))

    def release_sweep(self, after: Optional[torch.cuda.Event] = None) -> None:
        """Start the table sweep a forward-announced step held back (see _begin_overlapped); no-op otherwise.
        `after`: an event on the caller's stream the sweep is ordered behind, instead of everything queued so far."""
        pending, self._sweep_pending = getattr(self, "_sweep_pending", None), None
        self._hold_for_logits = False
        ops.held_sweeps.discard(self)
        if pending is not None:
            pending(True, after)

    # The sweep saturates HBM for as long as it lasts, and everything that runs next to it is stretched 2-3x.  When the
    # sweep IS the step (headline shape: 5.0 of 5.3 ms) that is free; when the forward/backward chain is as long as the
    # sweep or longer, a thinner sweep (fewer persistent workgroups) that ends with the chain instead of well before it
    # is faster overall -- C2: 1.27 ms at 768 workgroups, 1.14 at 512, 1.43 at 256; history model: best at 128-256.
    # The best level depends on the shapes, so it is MEASURED, on the events of steps that have already completed
    # (queried, never waited on: no host synchronisation is added):
    #   probe   full width until two steps have been seen; sweep > 0.9 of the step -> keep full width, done
    #   scan    otherwise the next len(levels) * SCAN_BLOCK steps run the levels one block each, enqueued OPEN LOOP (the
    #           host may be dozens of steps ahead of the GPU: waiting for each level's verdict before trying the next
    #           would stretch the scan over hundreds of steps)
    #   wait    full width until the last scan step's events are in, then the level with the fastest step is kept
    # and the whole thing repeats every RESCAN_STEPS steps.  Results do not depend on the level: the sweep's chunks
    # are handed out dynamically either way.
    _SWEEP_LEVELS = (0, 640, 512, 384, 256, 128)  # workgroups; 0 = library default (3 per CU = 768)
    _SCAN_BLOCK = 6  # steps per level; the first one of a block overlaps the previous level's tail and is not counted
    # row-sharded group: every level TWICE (down the list and back up), 16-step blocks, the first 4 steps of a block not
    # counted, the median of a level's 24 steps kept; the first scan of an optimiser starts after 40 steps.  Pinned levels
    # at the emulated W = 8 step (tools/bench_emulated_world.py, EMU_FORCE_LEVELS): 4.54 / 4.42 / 4.58 / 4.21 / 4.18 / 4.47
    # ms for 768 / 512 / 384 / 256 / 192 / 128 workgroups, the same within 0.02 ms at once after every switch.  But the
    # first ~150 steps of a process run 0.1 - 0.3 ms slower than the steady state and speed up as they go, so a one-way
    # scan right at the start read its LAST level (128) as the best and kept it: 4.5 instead of 4.2 ms (round 5).
    _GROUP_SCAN_BLOCK = 16
    _GROUP_SCAN_SKIP = 4
    _GROUP_SCAN_DELAY = 40
    _RESCAN_STEPS = 4000

    def sweep_level_note(self) -> str:
        """How the sweep's width was chosen (bench.py prints it)."""
        ts = self._tune_state
        if os.environ.get("TT_SWEEP_WGS") is not None:
            return "fixed by TT_SWEEP_WGS"
        if ts is None:
            return "not measured yet"
        return ts.get("why", ts["phase"])

    # Row-sharded group: the steps of all ranks are synchronised by the collectives, so the best level is a property of
    # the GROUP, and ranks that locked different levels on local timing noise would drag each other.  Same probe / scan /
    # decide as below, but every decision is taken at a step number all ranks reach, from the MAX over the ranks of each
    # level's best step time (two host waits per scan, i.e. per 4000 steps).
    def _tune_sweep_group(self) -> None:
        import torch.distributed as dist
        from . import collectives
        levels = self._SWEEP_LEVELS
        step = self._host_steps + 1  # the step about to be enqueued
        ts = self._tune_state
        if ts is None:  # (the first probe waits: see _GROUP_SCAN_DELAY)
            ts = self._tune_state = {"phase": "probe", "t0": step + self._GROUP_SCAN_DELAY, "since": 0}

        def collect():
            obs = {}
            for got in self._tune_done:
                # a step's time = begin to the NEXT step's begin where that is known: what lies between a step's end and
                # the next one's begin (routing the next batch's lookups) waits for more or less of the sweep's tail
                # depending on the level -- begin-to-end under-read the slowest level by 0.13 ms and picked it (round 5)
                last = got[6] if len(got) > 6 else got[3]
                last.synchronize()
                if got[5] < ts["t0"] + 4 or got[5] in ts.get("skip", ()):
                    continue
                obs.setdefault(got[4], []).append((got[0].elapsed_time(last), got[1].elapsed_time(got[2])))
            self._tune_done.clear()
            return obs

        def group(values, op):
            t = torch.tensor(values, dtype=torch.float32, device=self._hyper.device)
            collectives.all_reduce_(t, op=op)
            return [float(v) for v in t.cpu()]

        if ts["phase"] == "probe":
            self._sweep_wgs = levels[0]
            if step == ts["t0"] + 8:
                seen = collect().get(0, [])
                mine = 1.0 if (len(seen) >= 2 and all(sweep > 0.9 * st for st, sweep in seen[-2:])) else 0.0
                if group([mine], dist.ReduceOp.MIN)[0] > 0.5:
                    self._lock_sweep(ts, 0, "the sweep is the step on every rank")
                else:
                    plan, skip, nxt = [], set(), step
                    for lv in levels[1:] + levels[:0:-1]:  # down the levels and back up: a drift over the scan cancels
                        skip.update(range(nxt, nxt + self._GROUP_SCAN_SKIP))  # a block's first steps are not the level's steady state
                        plan += [lv] * self._GROUP_SCAN_BLOCK
                        nxt += self._GROUP_SCAN_BLOCK
                    skip.add(nxt)
                    ts.update(phase="scan", plan=plan, skip=skip, scan_start=step)
        if ts["phase"] == "scan":
            k = step - ts["scan_start"]
            self._sweep_wgs = ts["plan"][k] if k < len(ts["plan"]) else levels[0]
            if k == len(ts["plan"]) + 3:
                obs = collect()
                # the MEDIAN step of each level's block (the single-GPU controller keeps the fastest step; here one lucky step
                # decided between levels 7 % apart in steady state: emulated W = 8, 4.19 vs 4.51 ms, round 5)
                def median(v):
                    v = sorted(v)
                    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

                local = [median([st for st, _ in obs.get(lv, [])]) if len(obs.get(lv, [])) >= 3 else 1.0e9 for lv in levels]
                worst = group(local, dist.ReduceOp.MAX)
                best = min(range(len(levels)), key=lambda i: (worst[i], i))
                self._lock_sweep(ts, levels[best], "group (max over ranks of each level's median step time): "
                                 + str({(lv or 768): round(v, 3) for lv, v in zip(levels, worst) if v < 1.0e8}))
        elif ts["phase"] == "locked":
            self._tune_done.clear()
            ts["since"] += 1
            if ts["since"] >= self._RESCAN_STEPS:
                self._tune_state = {"phase": "probe", "t0": step, "since": 0}

    def _tune_sweep(self) -> None:
        if os.environ.get("TT_SWEEP_WGS") is not None:
            return
        if self._sharded and self._sharded[0]._tt_shard.world > 1:
            return self._tune_sweep_group()
        ts = self._tune_state
        if ts is None:
            ts = self._tune_state = {"phase": "probe", "obs": {}, "plan": [], "skip": set(), "last": 0, "since": 0}
        levels = self._SWEEP_LEVELS
        newest = 0
        while self._tune_done and self._tune_done[0][2].query() and self._tune_done[0][3].query():
            got = self._tune_done.pop(0)
            newest = got[5]
            if got[5] - int(getattr(self, "_resume_step", 0)) <= 4 or got[5] in ts["skip"]:
                continue  # the first steps SINCE CONSTRUCTION OR RESUME (allocations, first-use set-up) / the first step of a scan block
            ts["obs"].setdefault(got[4], []).append((got[0].elapsed_time(got[3]), got[1].elapsed_time(got[2])))
        if ts["phase"] == "probe":
            seen = ts["obs"].get(0, [])
            if len(seen) >= 2:
                if all(sweep > 0.9 * step for step, sweep in seen[-2:]):
                    self._lock_sweep(ts, 0, "the sweep is the step")
                else:
                    nxt = self._host_steps + 1  # the step about to be enqueued
                    for lv in levels[1:]:
                        ts["skip"].add(nxt)
                        ts["plan"] += [lv] * self._SCAN_BLOCK
                        nxt += self._SCAN_BLOCK
                    ts["skip"].add(nxt)  # back to full width: its first step overlaps the thinnest level's tail
                    ts["last"] = nxt - 1
                    ts["phase"] = "scan"
        if ts["phase"] == "scan":
            if ts["plan"]:
                self._sweep_wgs = ts["plan"].pop(0)
            else:
                self._sweep_wgs = levels[0]
                ts["phase"] = "wait"
        elif ts["phase"] == "wait":
            if newest >= ts["last"]:
                ms = {lv: min(step for step, _ in o) for lv, o in ts["obs"].items() if len(o) >= 2}
                self._lock_sweep(ts, min(ms, key=ms.get) if ms else 0, str({(lv or 768): round(v, 3) for lv, v in ms.items()}))
        elif ts["phase"] == "locked":
            ts["since"] += 1
            if ts["since"] >= self._RESCAN_STEPS:
                self._sweep_wgs = levels[0]
                ts.update(phase="probe", obs={}, plan=[], skip={self._host_steps + 1}, since=0)

    def _lock_sweep(self, ts: dict, level: int, why: str) -> None:
        self._sweep_wgs = level
        ts.update(phase="locked", obs={}, plan=[], skip=set(), since=0, why=f"{level or 768} workgroups: {why}")
        if os.environ.get("TT_TUNE_DEBUG"):
            print(f"[tt] sweep level: {level or 768} workgroups ({why})", file=sys.stderr)

    def _launch_plans(self, side: bool) -> None:
        """Enqueue the deferred row-plan sorts of a forward-mode step (see _begin_overlapped)."""
        if not self._plans_pending or self._begun is None:
            return
        self._plans_pending = False
        if not side:
            ops.RowPlan.build_many([ts.plan for ts in self._begun.values()])
            return
        if self._plan_stream is None:
            self._plan_stream = N.aux_stream(next(iter(self._begun)).device)
        self._plan_stream.wait_event(self._plan_ready)  # the id lists exist
        with torch.cuda.stream(self._plan_stream):
            ops.RowPlan.build_many([ts.plan for ts in self._begun.values()])
            self._plan_done = torch.cuda.Event()
            self._plan_done.record(self._plan_stream)

    def begin_step(self, lookups: Dict[torch.nn.Parameter, Sequence[torch.Tensor]], hold_sweep: bool = False) -> bool:
        """Forward-mode entry (called by the models' train_forward before any lookup): announce
        the id blocks per table, in the order the forward will look them up.  Returns False
        (and does nothing) unless ``overlap_sweep == "forward"`` applies."""
        if self.overlap_sweep != "forward" or not torch.is_grad_enabled():
            return False
        if torch.cuda.is_current_stream_capturing() and not self.capture_overlap:
            return False  # GraphedTrainStep sets capture_overlap: the side-stream sweep then becomes a branch of the graph
        if self._begun is not None:
            raise RuntimeError('overlap_sweep="forward" supports one train_forward per optimiser step')
        if not all(p.is_cuda for p in self._tables):
            return False
        for p in self._tables:
            p._tt_lookups.clear()
            p._tt_rowgrads.clear()
        mine = {id(p) for p in self._tables}
        self._begin_overlapped({id(p): [b.reshape(-1) for b in blocks] for p, blocks in lookups.items() if id(p) in mine},
                               hold_sweep=hold_sweep)
        return True

    # ------------------------------------------------------------------ replicated parameters of a row-sharded model
    def _start_dense_allreduce(self):
        """Every replicated parameter's gradient into ONE flat buffer (tt_pack_grads, one launch) and its all-reduce (SUM:
        each rank holds the gradient of ITS rows' share of the global-batch mean) started -- it travels underneath the
        table finish.  A parameter whose `.grad` is None is NOT updated -- like the single-device path and
        torch.optim.Adam -- and its slice of the buffer is cleared so that the layout stays what every rank expects.
        (Which parameters receive gradients must be the same on every rank, as under DistributedDataParallel without
        find_unused_parameters: it is a property of the model code, not of a rank's data.)"""
        from . import parallel
        lib = N.load()
        total = sum(p.numel() for p in self._dense)
        if total == 0:
            return None
        dev = self._dense[0].device
        if self._flat_g is None or self._flat_g.numel() != total:
            self._flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        descs = (N.AdamTensor * len(self._dense))()
        n, off, missing = 0, 0, []
        keep = []
        self._dense_live = [p.grad is not None for p in self._dense]
        for p in self._dense:
            if p.grad is not None:
                gr = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(gr)
                descs[n].p, descs[n].g, descs[n].n = self._flat_g.data_ptr() + 4 * off, gr.data_ptr(), p.numel()
                n += 1
            else:
                missing.append((off, p.numel()))
            off += p.numel()
        for o, k in missing:
            self._flat_g[o:o + k].zero_()
        if n:
            N.check(lib.tt_pack_grads(descs, n, N.stream()), "tt_pack_grads")
        return parallel.all_reduce_dense_start(self._flat_g), keep

    def _finish_dense_allreduce(self, reduce, lib, hyper) -> None:
        pending, _keep = reduce
        flat = pending.wait()
        descs = (N.AdamTensor * len(self._dense))()
        off, n = 0, 0
        for p, live in zip(self._dense, self._dense_live):
            if live:  # no gradient, no update (a frozen / unused parameter keeps its value AND its moments)
                st = self.state[p]
                descs[n].p, descs[n].g = p.data_ptr(), flat.data_ptr() + 4 * off
                descs[n].m, descs[n].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                descs[n].n = p.numel()
                n += 1
            off += p.numel()
        if n:
            N.check(lib.tt_adam_dense(descs, n, hyper, N.stream()), "tt_adam_dense")

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self._tables:
            p._tt_rowgrads.clear()
        super().zero_grad(set_to_none=set_to_none)
        if not getattr(self, "_hold_for_logits", False):
            self.release_sweep()
        if self._begun is not None and not torch.cuda.is_current_stream_capturing():
            self._launch_plans(side=True)
        if (self.overlap_sweep and self._begun is None and any(p._tt_lookups for p in self._tables)
                and self._tables[0].is_cuda and not torch.cuda.is_current_stream_capturing()):
            self._begin_overlapped()

    # ------------------------------------------------------------------ step
    @staticmethod
    def _ordered_rows(p) -> List[torch.Tensor]:
        """Gradient blocks in lookup order (backward delivers them in reverse)."""
        blocks = p._tt_rowgrads
        if len(blocks) != len(p._tt_lookups) or any(b.index is None for b in blocks):
            raise RuntimeError(
                "a table lookup of the last forward received no gradient (or more than one forward ran "
                "before zero_grad()); use DenseExactAdam(..., overlap_sweep=False) for this call pattern")
        return [b.rows for b in sorted(blocks, key=lambda b: b.index)]

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure is not supported")
        if not self._ready:
            self._init_state()
        lib = N.load()
        hyper = self._hyper.data_ptr()
        if self._sharded and self._begun is None:  # (checked BEFORE any collective is started)
            raise RuntimeError("row-sharded tables: step() without a train_forward that announced its lookups (the models' "
                               "train_forward does; a custom forward must call model._announce_lookups first)")
        reduce = self._start_dense_allreduce() if self._sharded else None
        self.release_sweep()
        if self._begun is not None:
            # overlapped schedule: hyper already advanced, tables already swept on the side stream
            self._launch_plans(side=False)  # nobody called zero_grad() after the forward: sort now, in line
            torch.cuda.current_stream().wait_event(self._sweep_done)
            if self._plan_done is not None:
                torch.cuda.current_stream().wait_event(self._plan_done)
                self._plan_done = None
            if self._begun:  # every table's looked-up rows in ONE launch
                jobs = (N.AdamFinishJob * len(self._begun))()
                for i, (p, ts) in enumerate(self._begun.items()):
                    st = self.state[p]
                    ts.plan.attach(self._ordered_rows(p))
                    j = jobs[i]
                    j.W, j.M, j.V = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    shard = getattr(p, "_tt_shard", None)
                    j.n_rows = p.shape[0] if shard is None else shard.n_local
                    j.dim, j.src, j.n_ids = p.shape[1], C.pointer(ts.plan.sources), ts.plan.n
                    j.sorted_ids, j.perm = ts.plan.sorted_ids.data_ptr(), ts.plan.perm.data_ptr()
                    j.seg_begin, j.n_unique = ts.plan.seg_begin.data_ptr(), ts.plan.n_unique.data_ptr()
                    j.side, j.side_bytes = (None, 0) if ts.marked else (ts.side.data_ptr(), ts.side.numel())
                N.check(lib.tt_adam_tables_finish(jobs, len(self._begun), hyper, N.stream()), "tt_adam_tables_finish")
            self._begun = None
            if self._tune is not None:
                end = torch.cuda.Event(enable_timing=True)
                end.record()
                self._tune[3] = end
                if len(self._tune_done) < 1024:  # never drop a PENDING measurement: when the host runs many steps ahead the
                    self._tune_done.append(self._tune)  # oldest one is the next to complete (new ones are skipped meanwhile)
                self._tune = None
        elif self.lazy:
            if self._prefetch_done is not None:  # rows being replayed for a later batch: finish first
                torch.cuda.current_stream().wait_event(self._prefetch_done)
                self._prefetch_done, self._prefetch_keep = None, None
            self._catchup_last.clear()
            self._advance_lazy()
            todo = []
            for p in self._tables:
                blocks = p._tt_rowgrads
                if p.grad is not None:
                    raise RuntimeError("embedding table received a dense gradient while in row-gradient mode")
                if not blocks:
                    continue
                if all(b.index is not None for b in blocks):
                    blocks = sorted(blocks, key=lambda b: b.index)
                todo.append((p, blocks, ops.RowPlan([b.ids for b in blocks], p.shape[0], slot=f"plan{len(todo)}", defer=True)))
            ops.RowPlan.build_many([plan for _, _, plan in todo])  # every table's sort in one launch where the lists are short
            for p, blocks, plan in todo:
                plan.attach([b.rows for b in blocks])
                st = self.state[p]
                n_rows, dim = p.shape
                wsp, wsn = ops._ws(p.device, lib.tt_adam_table_workspace_bytes(plan.n, dim), "adam_side")
                N.check(lib.tt_adam_table_lazy(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                               n_rows, dim, hyper, C.byref(plan.sources), plan.n,
                                               plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                               plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), wsp, wsn,
                                               self._last_step[id(p)].data_ptr(), self._tab.data_ptr(),
                                               self._tab_steps, N.stream()), "tt_adam_table_lazy")
        else:
            N.check(lib.tt_adam_advance(hyper, N.stream()), "tt_adam_advance")
            self._host_steps += 1
            for p in self._tables:
                blocks = p._tt_rowgrads
                st = self.state[p]
                n_rows, dim = p.shape
                if p.grad is not None:
                    raise RuntimeError("embedding table received a dense gradient while in row-gradient mode")
                if blocks:
                    if all(b.index is not None for b in blocks):
                        # forward (lookup) order, like the overlapped schedule: duplicates of a row
                        # across lookups are then summed in the same order by both schedules
                        blocks = sorted(blocks, key=lambda b: b.index)
                    plan = ops.RowPlan.from_grads(blocks, n_rows)
                    wsp, wsn = ops._ws(p.device, lib.tt_adam_table_workspace_bytes(plan.n, dim), "adam_side")
                    N.check(lib.tt_adam_table(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                              n_rows, dim, hyper, C.byref(plan.sources), plan.n,
                                              plan.sorted_ids.data_ptr(), plan.perm.data_ptr(),
                                              plan.seg_begin.data_ptr(), plan.n_unique.data_ptr(), wsp, wsn,
                                              N.stream()), "tt_adam_table")
                # a table with no lookups this step has grad None: torch.optim skips it too
        for p in self._tables:
            for b in p._tt_rowgrads:  # (a sharded table this rank owns no row of: its exchanges still have to be waited for)
                b.rows
            p._tt_lookups.clear()
            p._tt_rowgrads.clear()
            p._tt_active = None

        if reduce is not None:
            self._finish_dense_allreduce(reduce, lib, hyper)
            return None
        live = [p for p in self._dense if p.grad is not None]
        if live:
            descs = (N.AdamTensor * len(live))()
            keep = []
            for i, p in enumerate(live):
                gr = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(gr)
                st = self.state[p]
                descs[i].p, descs[i].g = p.data_ptr(), gr.data_ptr()
                descs[i].m, descs[i].v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                descs[i].n = p.numel()
            N.check(lib.tt_adam_dense(descs, len(live), hyper, N.stream()), "tt_adam_dense")
        return None
