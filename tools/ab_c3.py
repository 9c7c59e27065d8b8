"""A/B on ONE box, one process: per-step GPU time of a bench workload with a module-level switch off / on, alternating
blocks (events around every step, no host sync in the loop).
usage: python tools/ab_c3.py [workload] [module.attr] [steps per block] [blocks]     e.g.  C3 optim._HOLD_SWEEP 200 3"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import two_tower_models_amd as A  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
mod_name, attr = (sys.argv[2] if len(sys.argv) > 2 else "optim._HOLD_SWEEP").split(".")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 3
VALUES = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else None  # integer settings instead of off / on
mod = importlib.import_module("two_tower_models_amd." + mod_name)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg = dict(bench.WORKLOADS[wl])
batches = bench.make_batches(cfg, 16, dev)


def run(flag):
    setattr(mod, attr, flag)
    model = bench.build_model(cfg, dev)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")

    def step(i):
        loss = model.train_forward(*batches[i % 16])
        opt.zero_grad()
        loss.backward()
        opt.step()

    for i in range(120):
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
    opt_level[0] = opt.sweep_level_note()
    del model, opt
    torch.cuda.empty_cache()
    return ms[steps // 2], sum(ms) / steps, opt_level[0]


opt_level = [None]
for b in range(blocks):
    for flag in (VALUES or (False, True)):
        p50, mean, note = run(flag)
        print(f"{wl} {attr}={flag}: p50 {p50:.3f} ms, mean {mean:.3f} ms   [sweep: {note}]", flush=True)
