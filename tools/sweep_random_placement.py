"""How much does WHERE the six arrays of the P tables land change the sweep's rate on one box?  30 random placements
(every array at a random 2 MiB multiple inside its own 1 GiB of slack in one arena -- what a caching allocator's large
blocks look like) + torch's own six allocations, interleaved over rounds in one process.
    python tools/sweep_random_placement.py [trials]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N  # noqa: E402

lib = N.load()
dev = torch.device("cuda:0")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sizes = [1_000_000 * 128, 10_000_000 * 128]
G, M2 = 1 << 30, 2 << 20
regions = []
total = 0
for n in sizes:
    for _ in range(3):
        regions.append((total, n))
        total += (n * 4 + G + M2 - 1) // M2 * M2
# six separate allocations BEFORE the arena (the first allocations of the process, like a training script's), and six after
own_first = [[torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3)] for n in sizes]
arena = torch.empty(total // 4 + M2, dtype=torch.float32, device=dev)
arena.zero_()
a0 = (-arena.data_ptr()) % M2
own = [[torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(3)] for n in sizes]
hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 5, 0, 0, 0], dtype=torch.float64, device=dev)
N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "adv")


def arrays(offsets):
    out = []
    for t in range(2):
        row = []
        for k in range(3):
            start, n = regions[3 * t + k]
            b = (a0 + start + offsets[3 * t + k]) // 4
            row.append(arena[b:b + n])
        out.append(row)
    return out


def sweep(tabs, wgs, reps=4):
    d = (N.AdamTensor * 2)()
    for i, (p, m, v) in enumerate(tabs):
        d[i].p, d[i].m, d[i].v, d[i].n = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        N.check(lib.tt_adam_tables_sweep(d, 2, hyper.data_ptr(), wgs, N.stream()), "sweep")
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[1]


rng = random.Random(6)
cases = {"torch": None, "torch1st": "first"}
for i in range(trials):
    cases[f"rand{i:02d}"] = [rng.randrange(0, G // M2) * M2 for _ in range(6)]
res = {k: {512: [], 768: []} for k in cases}
for rnd in range(2):
    for name, offs in cases.items():
        tabs = own if offs is None else own_first if offs == "first" else arrays(offs)
        for w in (512, 768):
            res[name][w].append(sweep(tabs, w))
nb = 24.0 * sum(sizes)
rate = {k: {w: nb / min(v[w]) / 1e6 for w in (512, 768)} for k, v in res.items()}
print("placement: GB/s at 512 / 768 workgroups (best of 2 rounds x 2nd-fastest of 4)")
for name in cases:
    offs = cases[name]
    print(f"  {name:8s} {rate[name][512]:6.0f} {rate[name][768]:6.0f}  " + ("" if not isinstance(offs, list) else " ".join(f"{o // M2:4d}" for o in offs)))
for w in (512, 768):
    v = sorted(rate[k][w] for k in cases if k.startswith("rand"))
    print(f"{w} workgroups, {trials} random placements: min {v[0]:.0f}  p10 {v[len(v) // 10]:.0f}  median {v[len(v) // 2]:.0f}  max {v[-1]:.0f} GB/s "
          f"(spread {(v[-1] - v[0]) / v[-1] * 100:.1f} %); six separate torch allocations: made first in the process {rate['torch1st'][w]:.0f}, "
          f"made after the arena {rate['torch'][w]:.0f}")
