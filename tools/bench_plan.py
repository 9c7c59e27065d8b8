"""Time tt_rowgrad_plan alone (MI355X).  Usage: python tools/bench_plan.py [n n_rows]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import ops

dev = torch.device("cuda:0")
cases = [(4096, 1_000_000), (8192, 10_000_000), (213_000, 1_000_000)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for n, n_rows in cases:
    ids = torch.randint(0, n_rows, (n,), device=dev)
    plan = ops.RowPlan([ids], n_rows)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            plan.build()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20)
    print(f"n={n} n_rows={n_rows}: {sorted(ts)[2] * 1e3:.1f} us per plan "
          , flush=True)
