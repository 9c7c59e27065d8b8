"""Where do the ~0.42 ms between a step's start and its logits kernel go at the emulated W = 8 step?  HIP events around the
sub-phases of the sharded train_forward's front (lookups' routing, the optimiser's begin_step, item tower, all-gather start, user
tower) and the host's own clock for the same calls.   python tools/emu_front_probe.py [W]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import bench_emulated_world as emu  # noqa: E402
from two_tower_models_amd import collectives, parallel  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
real = (collectives.dist, parallel.dist)
collectives.dist = parallel.dist = emu._fake_dist(W)
cfg = dict(bench.WORKLOADS["P"])
model, opt = bench.build_sharded(cfg, dev, 0)
batches = bench.make_batches(cfg, 8, dev)
step = bench.sharded_step_fn(model, opt, torch.zeros((), device=dev))
for i in range(260):
    step(batches[i % 8], batches[(i + 1) % 8])
torch.cuda.synchronize()

marks, host = [], []


def wrap(obj, name, tag):
    fn = getattr(obj, name)

    def inner(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        t1 = time.perf_counter()
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        marks.append((tag, e0, e1)); host.append((tag, t0, t1))
        return out

    setattr(obj, name, inner)


wrap(parallel, "begin_lookups", "begin_lookups (route build, ids a2a, serve, rows a2a)")
wrap(opt, "begin_step", "opt.begin_step (park rows; sweep held)")
wrap(model, "compute_item_embeddings", "item tower")
wrap(parallel, "start_all_gather", "start_all_gather")
wrap(model, "compute_user_embedding", "user tower")
wrap(model, "compute_training_loss", "compute_training_loss (gather wait, logits fwd, loss)")
acc, hacc, n = {}, {}, 30
gap_acc = {}
for i in range(n):
    marks.clear(); host.clear()
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    h0 = time.perf_counter()
    step(batches[i % 8], batches[(i + 1) % 8])
    torch.cuda.synchronize()
    prev_e, prev_tag = s0, "step start"
    for (tag, e0, e1), (_, t0, t1) in zip(marks, host):
        acc[tag] = acc.get(tag, 0.0) + e0.elapsed_time(e1) / n
        hacc[tag] = hacc.get(tag, 0.0) + (t1 - t0) * 1e3 / n
        g = f"{prev_tag} -> {tag}"
        gap_acc[g] = gap_acc.get(g, 0.0) + prev_e.elapsed_time(e0) / n
        prev_e, prev_tag = e1, tag
print(f"emulated W = {W}, P shape: GPU ms between the events around each call (host ms of the call itself)")
for tag in acc:
    print(f"  {tag:64s} gpu {acc[tag]:7.3f}   host {hacc[tag]:7.3f}")
print("gaps on the GPU timeline between consecutive calls:")
for g, v in gap_acc.items():
    print(f"  {g:110s} {v:7.3f}")
collectives.dist, parallel.dist = real
