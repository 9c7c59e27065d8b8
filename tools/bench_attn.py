"""Time the encoder's attention kernels alone at the BASELINE shape (B = 4096, H = 50, 4 heads x 32) on MI355X.
The backward takes attention_wg.hip (one sample per workgroup) at the BASELINE shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N

lib = N.load()
dev = "cuda:0"
B, H, D, heads = 4096, 50, 128, 4
qkv = torch.randn(B * H, 3 * D, device=dev) * 0.5
ctx = torch.empty(B * H, D, device=dev)
lse = torch.empty(B, heads, H, device=dev)
d_ctx = torch.randn(B * H, D, device=dev)
d_qkv = torch.empty(B * H, 3 * D, device=dev)


def fwd():
    N.check(lib.tt_attn_fwd(qkv.data_ptr(), B, H, D, heads, ctx.data_ptr(), lse.data_ptr(), N.stream()), "tt_attn_fwd")


def bwd():
    N.check(lib.tt_attn_bwd(qkv.data_ptr(), ctx.data_ptr(), lse.data_ptr(), d_ctx.data_ptr(), B, H, D, heads, d_qkv.data_ptr(),
                            N.stream()), "tt_attn_bwd")


for name, fn, mf in (("fwd", fwd, 128), ("bwd", bwd, 312)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 100
    tf = B * heads * mf * 4096 / ms / 1e9
    print(f"attention {name}: {ms * 1e3:7.1f} us  ({mf} MFMAs per (sample, head) -> {tf:5.1f} TFLOP/s issued)")
