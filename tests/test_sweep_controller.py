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
    me = types.SimpleNamespace(_tune_done=[], _tune_state=None, _host_steps=0, _sweep_wgs=0, _sharded=[],
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


def _drive_group(local_ms, sweep_ms, n_steps, monkeypatch):
    """Two simulated ranks of a row-sharded group run DenseExactAdam._tune_sweep_group in lockstep (one thread each; the
    all-reduce is a barrier + a shared mailbox).  local_ms[r][level] = rank r's step time at that sweep level."""
    import threading

    import torch
    import torch.distributed as dist

    from two_tower_models_amd import collectives

    world = len(local_ms)
    barrier, mailbox = threading.Barrier(world), {}
    rank_of = threading.local()

    def fake_all_reduce(t, op=None):
        mailbox[rank_of.r] = t.clone()
        barrier.wait()
        stacked = torch.stack([mailbox[r] for r in range(world)])
        out = stacked.min(0).values if op == dist.ReduceOp.MIN else stacked.max(0).values
        barrier.wait()
        t.copy_(out)
        return t

    monkeypatch.setattr(collectives, "all_reduce_", fake_all_reduce)
    hist, mes, errors = [[] for _ in range(world)], [None] * world, []

    def run(r):
        try:
            rank_of.r = r
            me = types.SimpleNamespace(_tune_done=[], _tune_state=None, _host_steps=0, _sweep_wgs=0,
                                       _hyper=types.SimpleNamespace(device="cpu"),
                                       _SWEEP_LEVELS=DenseExactAdam._SWEEP_LEVELS, _SCAN_BLOCK=DenseExactAdam._SCAN_BLOCK,
                                       _GROUP_SCAN_BLOCK=DenseExactAdam._GROUP_SCAN_BLOCK, _GROUP_SCAN_SKIP=DenseExactAdam._GROUP_SCAN_SKIP, _GROUP_SCAN_DELAY=DenseExactAdam._GROUP_SCAN_DELAY, _RESCAN_STEPS=DenseExactAdam._RESCAN_STEPS)
            me._lock_sweep = types.MethodType(DenseExactAdam._lock_sweep, me)
            tune = types.MethodType(DenseExactAdam._tune_sweep_group, me)
            mes[r] = me
            for _ in range(n_steps):
                tune()
                me._host_steps += 1
                lv = me._sweep_wgs
                ev = lambda t: types.SimpleNamespace(t=t, synchronize=lambda: None, elapsed_time=lambda other, t=t: other.t - t)
                me._tune_done.append([ev(0.0), ev(0.0), ev(sweep_ms), ev(local_ms[r][lv]), lv, me._host_steps])
                hist[r].append(lv)
        except Exception as e:  # a dead thread must not leave the other one waiting on the barrier
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not errors, errors
    return mes, hist


def test_group_tuner_takes_the_level_whose_worst_rank_is_fastest_at_the_same_step_on_every_rank(monkeypatch):
    """Row-sharded group (optim.py::_tune_sweep_group): the sweep level is chosen ONCE for the group, at a step number
    every rank reaches (ranks locking different levels on local timing noise drag each other).  Two ranks whose
    local measurements favour different levels end on the level whose MAX over ranks is smallest, and hold the same level
    at every step, including the step the decision lands on."""
    local_ms = [{0: 4.40, 640: 4.38, 512: 4.20, 384: 4.45, 256: 4.60, 128: 5.0},   # rank 0 alone would take 512
                {0: 4.50, 640: 4.45, 512: 4.70, 384: 4.30, 256: 4.40, 128: 5.2}]   # rank 1 alone 384; group (max): 640
    mes, hist = _drive_group(local_ms, 1.2, 400, monkeypatch)
    assert hist[0] == hist[1]
    assert mes[0]._sweep_wgs == mes[1]._sweep_wgs == 640
    assert mes[0]._tune_state["phase"] == "locked" and "group" in mes[0]._tune_state["why"]
    for lv in (512, 384, 256, 128):  # every level exactly two blocks while scanning
        assert hist[0].count(lv) == 2 * DenseExactAdam._GROUP_SCAN_BLOCK  # down the levels and back up


def test_group_tuner_keeps_full_width_only_if_the_sweep_is_the_step_on_every_rank(monkeypatch):
    sweep = 5.0
    bound = {lv: 5.3 for lv in DenseExactAdam._SWEEP_LEVELS}
    mes, hist = _drive_group([bound, bound], sweep, 200, monkeypatch)
    assert set(hist[0]) == {0} and mes[0]._tune_state["phase"] == "locked"
    # one rank whose step is much longer than its sweep: the group scans
    chain = {0: 7.0, 640: 6.9, 512: 6.5, 384: 6.6, 256: 6.8, 128: 7.5}
    mes, hist = _drive_group([bound, chain], sweep, 400, monkeypatch)
    assert hist[0] == hist[1] and mes[0]._sweep_wgs == 512
