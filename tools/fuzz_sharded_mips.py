"""Randomised check of the row-sharded MIPS (per-rank top-K, all-to-all of the candidates, tt_mips_merge) with the
product backend at world sizes 2-4 on one GPU (gloo): random corpus sizes (down to fewer rows than ranks x K) and K,
indices and scores against the unsharded exact order (tests/test_gpu_parallel.py's assertion).
    python tools/fuzz_sharded_mips.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import test_gpu_parallel as T  # noqa: E402

if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        world = int(rng.choice([2, 3, 4]))
        C = int(rng.choice([world, world + 1, 17, 100, 999, 5000, 20000, 70001]))
        K = int(rng.integers(1, min(C, 600) + 1))
        try:
            T.test_multi_rank_sharded_mips(world, C, K, "gloo")
        except BaseException as e:  # noqa: BLE001
            bad += 1
            print(f"FINDING case {n}: W={world} C={C} K={K} | {type(e).__name__} {str(e)[:400]}", flush=True)
        n += 1
    print(f"{n} cases, {bad} findings in {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
