"""Step time of the default bench workload as a function of time since the process started using the GPU:
windows of 50 steps, printed with the elapsed wall time.  (Are the first seconds on a fresh box slower?)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import two_tower_models_amd as A  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
t_start = time.perf_counter()
cfg = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "P"])
model = bench.build_model(cfg, dev)
opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
batches = bench.make_batches(cfg, 16, dev)
total = torch.zeros((), device=dev)
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for w in range(windows):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50):
        loss = model.train_forward(*batches[i % 16])
        opt.zero_grad()
        loss.backward()
        opt.step()
        total.add_(loss.detach())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"t={time.perf_counter() - t_start:6.1f}s window {w:3d}: {dt / 50 * 1e3:.3f} ms/step", flush=True)
