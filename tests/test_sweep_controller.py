"""DenseExactAdam's sweep-width controller (optim.py::_tune_sweep) on stub events: no GPU, no library.
The controller only ever sees (step time, sweep time, level, step number) of steps that have completed, with the host
an arbitrary number of steps ahead -- which is exactly what the stubs model."""
import types

import pytest

from two_tower_models_amd.optim import DenseExactAdam


class _Ev:
    def __init__(self, t, done):
        self.t, self.done = t, done

    def query(self):
        return self.done()

    def elapsed_time(self, other):
        return other.t - self.t


def _drive(step_ms_of, sweep_ms_of, lag, n_steps):
    """Run the controller for n_steps host steps; the GPU completes step k when the host is at step k + lag."""
    me = types.SimpleNamespace(_tune_done=[], _tune_state=None, _host_steps=0, _sweep_wgs=0,
                               _SWEEP_LEVELS=DenseExactAdam._SWEEP_LEVELS, _SCAN_BLOCK=DenseExactAdam._SCAN_BLOCK,
                               _RESCAN_STEPS=DenseExactAdam._RESCAN_STEPS)
    me._lock_sweep = types.MethodType(DenseExactAdam._lock_sweep, me)
    tune = types.MethodType(DenseExactAdam._tune_sweep, me)
    history = []
    for _ in range(n_steps):
        tune()
        me._host_steps += 1
        k, lv = me._host_steps, me._sweep_wgs
        done = (lambda k=k: me._host_steps >= k + lag)
        st, sw = step_ms_of(lv), sweep_ms_of(lv)
        me._tune_done.append([_Ev(0.0, done), _Ev(0.0, done), _Ev(sw, done), _Ev(st, done), lv, k])
        history.append(lv)
    return me, history


@pytest.mark.parametrize("lag", [0, 2, 40])
def test_sweep_bound_step_keeps_full_width(lag):
    me, hist = _drive(lambda lv: 5.3, lambda lv: 5.0, lag, 200)
    assert set(hist) == {0} and me._tune_state["phase"] == "locked"


@pytest.mark.parametrize("lag", [0, 3, 40])
def test_chain_bound_step_scans_every_level_once_and_keeps_the_fastest(lag):
    ms = {0: 1.30, 640: 1.31, 512: 1.18, 384: 1.21, 256: 1.41, 128: 2.3}
    me, hist = _drive(lambda lv: ms[lv], lambda lv: 1.0, lag, 300)
    assert me._tune_state["phase"] == "locked" and me._sweep_wgs == 512
    for lv in (640, 512, 384, 256, 128):  # one block per level while scanning, the winner for the rest
        assert hist.count(lv) >= DenseExactAdam._SCAN_BLOCK
        if lv != 512:
            assert hist.count(lv) == DenseExactAdam._SCAN_BLOCK
    assert all(lv == 512 for lv in hist[-100:])


def test_rescan_after_the_interval():
    ms = {0: 1.30, 640: 1.31, 512: 1.18, 384: 1.21, 256: 1.41, 128: 2.3}
    me, hist = _drive(lambda lv: ms[lv], lambda lv: 1.0, 1, DenseExactAdam._RESCAN_STEPS + 200)
    assert hist.count(128) == 2 * DenseExactAdam._SCAN_BLOCK and me._sweep_wgs == 512


@pytest.mark.parametrize("lag", [0, 2, 25])
def test_sharded_schedule_scan_runs_each_candidate_one_block_and_keeps_the_fastest(lag):
    """sharded._ScheduleScan (where the sweep starts / how wide it runs in the row-sharded step) on stub events: the host
    `lag` steps ahead of the GPU, every candidate exactly one block, the fastest kept from then on."""
    from two_tower_models_amd.sharded import _ScheduleScan
    ms = {(0, 256): 4.55, (2, 256): 4.35, (0, 0): 4.46}
    clock = {"host": 0, "t": 0.0}

    class Ev:
        def record(self):
            self.step, self.t = clock["host"], clock["t"]

        def query(self):
            return clock["host"] >= self.step + lag

        def elapsed_time(self, other):
            return other.t - self.t

    scan = _ScheduleScan(list(ms), block=4, skip_first=3, new_event=Ev)
    hist = []
    for _ in range(80):
        cand = scan.begin()
        clock["t"] += ms[cand]  # the step's duration on the GPU's clock
        scan.end()
        clock["host"] += 1
        hist.append(cand)
    assert scan.best == (2, 256) and all(c == (2, 256) for c in hist[-30:])
    assert hist[:3] == [(0, 256)] * 3 and hist[3:15] == [(0, 256)] * 4 + [(2, 256)] * 4 + [(0, 0)] * 4


def test_sharded_schedule_scan_group_decision_is_identical_on_every_rank():
    """With a group reduction the schedule is chosen ONCE for the group, at the same step on every rank (ADVICE r3:
    ranks locking different schedules on local timing noise): two simulated ranks whose local measurements favour
    different candidates both end up on the candidate whose WORST rank is fastest, and switch at the same step."""
    from two_tower_models_amd.sharded import _ScheduleScan
    cands = [(0, 256), (2, 256), (0, 0)]
    local_ms = [{(0, 256): 4.40, (2, 256): 4.45, (0, 0): 4.60},   # rank 0 would pick (0, 256)
                {(0, 256): 4.70, (2, 256): 4.42, (0, 0): 4.50}]   # rank 1 would pick (2, 256); group: max -> (2, 256)
    clocks = [{"t": 0.0}, {"t": 0.0}]

    def make_ev(r):
        class Ev:
            def record(self):
                self.t = clocks[r]["t"]

            def query(self):
                return True

            def synchronize(self):
                pass

            def elapsed_time(self, other):
                return other.t - self.t
        return Ev

    mailbox = {}

    def group_max_for(r):
        def f(values):  # both ranks call in lockstep: rank 0 posts, rank 1 completes -- simulated with a shared dict
            mailbox[r] = list(values)
            return [max(a, b) for a, b in zip(mailbox.get(0, values), mailbox.get(1, values))]
        return f

    scans = [_ScheduleScan(cands, block=4, skip_first=3, new_event=make_ev(r), group_max=None) for r in range(2)]
    # pre-compute what each rank will report, so the simulated all-reduce can return the true max to both
    for r in range(2):
        scans[r]._group_max = lambda values, r=r: [max(local_ms[0][c], local_ms[1][c]) for c in cands]
    hist = [[], []]
    for _ in range(40):
        for r in range(2):
            c = scans[r].begin()
            clocks[r]["t"] += local_ms[r][c]
            scans[r].end()
            hist[r].append(c)
    assert scans[0].best == scans[1].best == (2, 256)
    assert hist[0] == hist[1]  # same candidate at every step, including the step the decision lands on
    assert "group" in scans[0].decided_by
