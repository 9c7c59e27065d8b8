"""Serving latency of BaselineMIPSModule.search for small query batches (C = 10 M, D = 128, K = 1000, bf16 and fp32
storage), back-to-back calls.  (A hipGraph replay of the ~20 launches of a call was measured with this script and
changes nothing -- 0.705 vs 0.714 ms at B = 16 bf16: the calls are asynchronous and the corpus stream, 2.56 GB at
~3.9 TB/s, is the time.)
    python tools/bench_mips_latency.py [C] [K]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import two_tower_models_amd as A

Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
D, dev = 128, "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
m = A.BaselineMIPSModule(corpus_size=8, embedding_dim=D)
m.corpus = torch.randn(Cn, D, device=dev, generator=g)
m.corpus_size = Cn


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name in ("fp32", "bf16"):
    if name == "bf16":
        m.use_bf16_storage()
    for B in (1, 16, 64, 256):
        q = torch.randn(B, D, device=dev, generator=g)
        ms = timed(lambda: m.search(q, K))
        print(f"{name} B={B:4d}: {ms:7.3f} ms per call  ({B / ms:.1f} K queries/s)", flush=True)
