"""Stand-alone timing of the fused encoder layer forward (tt_enc_layer_fwd) vs the three launches it replaces."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_tower_models_amd import _native as N, ops
lib = N.load()
B, H, D, heads = 4096, 50, 128, 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(B * H, D, device=dev)
w_in = torch.randn(3 * D, D, device=dev) * 0.09
b_in = torch.randn(3 * D, device=dev) * 0.1
w_out = torch.randn(D, D, device=dev) * 0.09
b_out = torch.randn(D, device=dev) * 0.1
y = torch.empty(B * H, D, device=dev)
qkv = torch.empty(B * H, 3 * D, device=dev)
ctx = torch.empty(B * H, D, device=dev)
lse = torch.empty(B, heads, H, device=dev)

def fused(save=True):
    N.check(lib.tt_enc_layer_fwd(x.data_ptr(), B, H, D, heads, w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(), b_out.data_ptr(),
                                 y.data_ptr(), D, 0, qkv.data_ptr() if save else None, ctx.data_ptr() if save else None,
                                 lse.data_ptr() if save else None, N.stream()), "fused")

def unfused():
    ops.gemm(N.TT_GEMM_NT, x, w_in, qkv, B * H, 3 * D, D, bias=b_in)
    c, l = ops._attn_fwd(qkv, B, H, D, heads)
    ops.gemm(N.TT_GEMM_NT, c, w_out, y, B * H, D, D, bias=b_out)

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

flop = 2 * B * H * D * 3 * D + 4 * B * heads * H * H * (D // heads) + 2 * B * H * D * D
for name, fn in (("fused (with by-products)", fused), ("fused (no by-products)", lambda: fused(False)), ("three launches", unfused)):
    t = timeit(fn)
    print(f"{name:28s} {t:8.1f} us   {flop / t / 1e6:6.1f} TFLOP/s (unpadded work)")
