"""A/B the Adam table sweep variants inside ONE process (interleaved rounds), MI355X.
    python tools/bench_sweep.py            # 10 M x 128 table"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    from two_tower_models_amd import _native as N
    lib = N.load()
    rows, dim = 10_000_000, 128
    dev = torch.device("cuda:0")
    W = torch.randn(rows, dim, device=dev)
    M = torch.randn(rows, dim, device=dev) * 0.01
    V = torch.rand(rows, dim, device=dev) * 0.01
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 5, 0, 0, 0], dtype=torch.float64, device=dev)
    N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "adv")
    def run():
        N.check(lib.tt_adam_table(W.data_ptr(), M.data_ptr(), V.data_ptr(), rows, dim, hyper.data_ptr(), None, 0,
                                  None, None, None, None, None, 0, N.stream()), "adam")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            run()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 4)
    ms = sorted(ts)[len(ts) // 2]
    print(f"variant={os.environ.get('TT_SWEEP_VARIANT')} bpc={os.environ.get('TT_SWEEP_BLOCKS_PER_CU')} "
          f"median {ms:.3f} ms  min {min(ts):.3f} ms  {24.0 * rows * dim / ms / 1e6:.0f} GB/s", flush=True)
else:
    for rnd in range(2):
        for variant in (0, 1, 2, 3):
            for bpc in (4, 8, 16):
                env = dict(os.environ, TT_SWEEP_VARIANT=str(variant), TT_SWEEP_BLOCKS_PER_CU=str(bpc))
                subprocess.run([sys.executable, __file__, "--one"], env=env, stdin=subprocess.DEVNULL)
