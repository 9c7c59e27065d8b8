"""Where does the HOST spend a row-sharded step (emulated world size, stand-in collectives)?  cProfile over the enqueue loop:
a call that blocks on the GPU shows up with milliseconds of tottime.     python tools/host_profile_sharded.py [W] [workload]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import bench_emulated_world as E  # noqa: E402
from two_tower_models_amd import sharded  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "P"
dev = torch.device("cuda:0")
sharded.dist = E._fake_dist(W)
tr = sharded.ShardedTrainer(dict(bench.WORKLOADS[wl]), dev, negatives="global")
b = tr.make_batches(8)
for i in range(30):
    tr.step(b[i % 8], b[(i + 1) % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40):
    tr.step(b[i % 8], b[(i + 1) % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"W={W} {wl}: host enqueue {1e3 * (t1 - t0) / 40:.3f} ms/step, step {1e3 * (t2 - t0) / 40:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for i in range(40):
    tr.step(b[i % 8], b[(i + 1) % 8])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
