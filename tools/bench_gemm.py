"""Time tt_gemm_f32 on the encoder-projection shapes (MI355X)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N
from two_tower_models_amd import ops

N.load()
dev = "cuda:0"
cases = [("NT qkv ", 0, 204800, 384, 128), ("NT out ", 0, 204800, 128, 128), ("NN dx  ", 1, 204800, 128, 384),
         ("NN dctx", 1, 204800, 128, 128), ("TN dWin", 2, 384, 128, 204800), ("TN dWkv", 2, 256, 128, 204800), ("TN dWo ", 2, 128, 128, 204800),
         ("NT twr ", 0, 8192, 128, 256)]
only = os.environ.get("TT_BENCH_ONLY")
for name, layout, M, Nn, K in cases:
    if only and only not in name:
        continue
    A = torch.randn((M, K) if layout != 2 else (K, M), device=dev)
    B = torch.randn((Nn, K) if layout == 0 else (K, Nn), device=dev)
    out = torch.empty(M, Nn, device=dev)
    bias = torch.randn(Nn, device=dev)
    if layout == 2:
        bias = None  # weight gradients carry no bias (the streaming TN path)
    for _ in range(30):
        ops.gemm(layout, A, B, out, M, Nn, K, bias=bias)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        ops.gemm(layout, A, B, out, M, Nn, K, bias=bias)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 100
    print(f"{name} M={M} N={Nn} K={K}: {ms * 1e3:8.1f} us  {2.0 * M * Nn * K / ms / 1e9:6.1f} TFLOP/s "
          , flush=True)
