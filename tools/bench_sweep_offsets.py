"""Does the relative placement of the p / m / v arrays matter for the Adam sweep?
One process, interleaved rounds; arrays are views at chosen byte offsets of one big buffer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N

lib = N.load()
rows, dim = 10_000_000, 128
n = rows * dim
dev = torch.device("cuda:0")
PAD = 64 << 20
big = torch.empty(3 * (n * 4 + PAD) // 4 + 1024, dtype=torch.float32, device=dev)
base = big.data_ptr()
print("base address alignment:", hex(base % (1 << 30)))
hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 5, 0, 0, 0], dtype=torch.float64, device=dev)
N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "adv")


def views(off_m, off_v):
    """p at 0, m at n*4 + off_m, v at 2*n*4 + off_v (bytes, multiples of 16)."""
    e = lambda byte: byte // 4
    W = big[0:n]
    M = big[e(n * 4 + off_m): e(n * 4 + off_m) + n]
    V = big[e(2 * n * 4 + off_v): e(2 * n * 4 + off_v) + n]
    return W, M, V


def run(W, M, V, reps=4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        N.check(lib.tt_adam_table(W.data_ptr(), M.data_ptr(), V.data_ptr(), rows, dim, hyper.data_ptr(), None, 0,
                                  None, None, None, None, None, 0, N.stream()), "adam")
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


big.normal_()
big.abs_().mul_(0.01)
cases = [(0, 0), (256, 512), (4096 + 256, 8192 + 512), (65536 + 4096, 2 * 65536 + 8192), (1 << 20, 2 << 20),
         ((1 << 20) + 4096 + 256, (2 << 20) + 8192 + 512), (16 << 20, 32 << 20)]
res = {c: [] for c in cases}
for rnd in range(4):
    for c in cases:
        W, M, V = views(*c)
        run(W, M, V, 1)
        res[c].append(run(W, M, V))
for c in cases:
    ts = sorted(res[c])
    print(f"offsets m+{c[0]:>9d} v+{c[1]:>9d}: median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  "
          f"{24.0 * n / ts[len(ts)//2] / 1e6:.0f} GB/s")
# reference points: plain float4 copy of the same total bytes (read n*4*3, write n*4*3)
src, dst = big[0:3 * n // 2], big[3 * n // 2: 3 * n]
for _ in range(2):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 4
    print(f"torch copy_ of {src.numel()*4/1e9:.1f} GB: {ms:.3f} ms = {2*src.numel()*4/ms/1e6:.0f} GB/s (read+write)")
