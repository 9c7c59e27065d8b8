"""Which RELATIVE placements of the p / m / v arrays slow the Adam sweep down?  (r6 box A: all six arrays on 1 GiB
boundaries -3.6 % against torch's own placement.)  The item table only (10 M x 128), p on a 1 GiB boundary, m and v on
their own 1 GiB boundaries PLUS an offset; one process, interleaved rounds.
    python tools/sweep_offsets2.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N  # noqa: E402

lib = N.load()
dev = torch.device("cuda:0")
n = 10_000_000 * 128
G = 1 << 30
big = torch.empty((3 * (n * 4 + 2 * G) + 2 * G) // 4, dtype=torch.float32, device=dev)
big.zero_()
base = big.data_ptr()
a0 = (-base) % G
span = (n * 4 + G - 1) // G * G + G  # distance between the arrays' 1 GiB anchors


def view(byte_off):
    return big[(a0 + byte_off) // 4:(a0 + byte_off) // 4 + n]


hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 5, 0, 0, 0], dtype=torch.float64, device=dev)
N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "adv")


def run(om, ov, wgs, reps=5):
    p, m, v = view(0), view(span + om), view(2 * span + ov)
    d = (N.AdamTensor * 1)()
    d[0].p, d[0].m, d[0].v, d[0].n = p.data_ptr(), m.data_ptr(), v.data_ptr(), n
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        N.check(lib.tt_adam_tables_sweep(d, 1, hyper.data_ptr(), wgs, N.stream()), "sweep")
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]


M = 1 << 20
cases = [(0, 0)] + [(k * M, 2 * k * M) for k in (2, 4, 8, 16, 32, 64, 128, 256)] + \
        [(k * M, 0) for k in (2, 16, 128, 512)] + [(6 * M, 10 * M), (22 * M, 50 * M), (170 * M, 342 * M), (682 * M, 346 * M),
                                                    (4096 + 2 * M, 8192 + 6 * M), (M // 2, M), (M // 8, M // 4)]
res = {}
for rnd in range(3):
    for c in cases:
        for w in (512, 768):
            res.setdefault((c, w), []).append(run(c[0], c[1], w))
nb = 24.0 * n
print("offsets of m / v beyond their 1 GiB anchors (MiB) -> GB/s at 512 / 768 workgroups (median of 3 rounds x median of 5)")
for c in cases:
    row = []
    for w in (512, 768):
        v = sorted(res[(c, w)])
        row.append(nb / v[1] / 1e6)
    print(f"  m +{c[0] / M:9.3f}  v +{c[1] / M:9.3f}:  {row[0]:6.0f}  {row[1]:6.0f}")
