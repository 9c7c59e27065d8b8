"""Time the encoder's collapsed last layer alone (csrc/encoder_last.hip: 3 forward + 4 backward launches) at the BASELINE
shape, B = 4096, H = 50, D = 128, 4 heads -- against its in-step times (profiles/r05_timeline_C3.txt) and the HBM floor
(x read once forward: 105 MB; x read + dx written backward: 210 MB).      python tools/bench_enc_last.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from two_tower_models_amd import _native as N

lib = N.load()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H, D, heads = 50, 128, 4
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.3
x, w_in, b_in, w_out, b_out = r(B * H, D), r(3 * D, D), r(3 * D), r(D, D), r(D)
out, d_recent = torch.empty(B, 2 * D, device=dev), r(B, 2 * D)
q0, tq, probs = torch.empty(B, D, device=dev), torch.empty(B, heads, D, device=dev), torch.empty(B, heads, H, device=dev)
xbar, ctx0 = torch.empty(B, heads, D, device=dev), torch.empty(B, D, device=dev)
dx, dW_in, db_in = torch.empty(B * H, D, device=dev), torch.empty(3 * D, D, device=dev), torch.empty(3 * D, device=dev)
dW_out, db_out = torch.empty(D, D, device=dev), torch.empty(D, device=dev)
wsn = lib.tt_enc_last_bwd_workspace_bytes(B, H, D, heads)
ws = torch.empty(wsn, dtype=torch.uint8, device=dev)


def fwd():
    N.check(lib.tt_enc_last_fwd(x.data_ptr(), B, H, D, heads, w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(), b_out.data_ptr(),
                                out.data_ptr(), 2 * D, q0.data_ptr(), tq.data_ptr(), probs.data_ptr(), xbar.data_ptr(),
                                ctx0.data_ptr(), N.stream()), "tt_enc_last_fwd")


def bwd():
    N.check(lib.tt_enc_last_bwd(x.data_ptr(), B, H, D, heads, w_in.data_ptr(), w_out.data_ptr(), d_recent.data_ptr(), 2 * D,
                                q0.data_ptr(), tq.data_ptr(), probs.data_ptr(), xbar.data_ptr(), ctx0.data_ptr(), dx.data_ptr(),
                                dW_in.data_ptr(), db_in.data_ptr(), dW_out.data_ptr(), db_out.data_ptr(), ws.data_ptr(), wsn,
                                N.stream()), "tt_enc_last_bwd")


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tf, tb = timed(fwd), timed(bwd)
mb = B * H * D * 4 / 1e6
print(f"B={B}: forward (pre + main + post) {tf:.1f} us  [x read once: {mb:.0f} MB = {mb / 6.4:.1f} us at 6.4 TB/s]")
print(f"        backward (a + main + b + reduce) {tb:.1f} us  [x read + dx written: {2 * mb:.0f} MB = {2 * mb / 6.4:.1f} us]")
