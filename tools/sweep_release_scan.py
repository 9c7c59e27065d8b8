"""Where in the step should a MARKED table sweep start?  (optim._MARK_ROWS: the sweep steps over the step's looked-up rows, so
nothing orders it against the forward's lookups any more.)  C3 on one box, one process: the sweep is released behind the
n-th occurrence of a named launch on the main stream; per setting the step time at the controller's chosen width and the
controller's whole level table.     usage: python tools/sweep_release_scan.py [steps] [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import two_tower_models_amd as A  # noqa: E402
from two_tower_models_amd import _native as N, ops, optim  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
wl = sys.argv[2] if len(sys.argv) > 2 else "C3"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = dict(bench.WORKLOADS[wl])
batches = bench.make_batches(cfg, 16, dev)
POINTS = [("parked rows, the library's release point", None, 0), ("tt_hist_embed_pool", "tt_hist_embed_pool", 1), ("in-proj 0", "tt_gemm_f32", 5), ("attn fwd 0", "tt_attn_fwd", 1),
          ("in-proj 1", "tt_gemm_f32", 6), ("attn fwd 1", "tt_attn_fwd", 2), ("enc_last fwd", "tt_enc_last_fwd", 1),
          ("user tower fwd", "tt_tower_fwd_x", 1), ("ce fwd", "tt_inbatch_ce_fwd_du_loss", 1), ("ce bwd", "tt_inbatch_ce_bwd", 1),
          ("enc_last bwd", "tt_enc_last_bwd_data", 1), ("attn bwd 1", "tt_attn_bwd", 1), ("attn bwd 0", "tt_attn_bwd", 2)]
real_check = N.check
state = {"opt": None, "name": None, "occ": 0, "seen": 0, "busy": False, "in_step": False}
real_release = A.DenseExactAdam.release_sweep


def release(self, after=None):  # the library's own release points are ignored while a scan point is set (step() is the last resort)
    if state["name"] is None or state["busy"] or state["in_step"]:
        real_release(self, after)


A.DenseExactAdam.release_sweep = release


def check(rc, what):
    real_check(rc, what)
    if state["name"] == what and not state["busy"]:
        state["seen"] += 1
        if state["seen"] == state["occ"]:
            state["busy"] = True
            ev = torch.cuda.Event()
            ev.record()
            state["opt"].release_sweep(after=ev)
            state["busy"] = False


N.check = check


def run(label, name, occ):
    optim._MARK_ROWS = name is not None
    model = bench.build_model(cfg, dev)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    state.update(opt=opt, name=name, occ=occ)

    def step(i):
        state["seen"] = 0
        loss = model.train_forward(*batches[i % 16])
        opt.zero_grad()
        loss.backward()
        state["in_step"] = True
        opt.step()
        state["in_step"] = False

    for i in range(200):
        step(i)
    torch.cuda.synchronize()
    evs = []
    for i in range(steps):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        step(i)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    torch.cuda.synchronize()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    note = opt.sweep_level_note()
    state["opt"] = None
    del model, opt
    torch.cuda.empty_cache()
    print(f"{wl} sweep released after {label}: p50 {ms[steps // 2]:.3f} ms   [{note}]", flush=True)


only = os.environ.get("SCAN_ONLY")  # one release point (its label), e.g. under rocprofv3 for a timeline
if only:
    run(*[pt for pt in POINTS if pt[0] == only][0])
    sys.exit(0)
for label, name, occ in POINTS:
    run(label, name, occ)
run(*POINTS[0])
optim._MARK_ROWS = True
state["name"] = None
run_default = POINTS[0][0].replace("parked", "marked")
# the shipped schedule: marked rows, the library's own release point
def shipped():
    model = bench.build_model(cfg, dev)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    evs = []
    for i in range(200 + steps):
        if i >= 200:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        loss = model.train_forward(*batches[i % 16])
        opt.zero_grad()
        loss.backward()
        opt.step()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
    torch.cuda.synchronize()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    print(f"{wl} {run_default}: p50 {ms[steps // 2]:.3f} ms   [{opt.sweep_level_note()}]", flush=True)


shipped()
