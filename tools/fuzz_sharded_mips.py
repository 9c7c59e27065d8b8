"""Randomised check of the row-sharded BaselineMIPSModule behind TwoTowerWithDebiasing.forward() (per-rank top-K,
all-to-all of the candidates, tt_mips_merge, rows fetched from their owners) with the product kernels at world sizes
2-4 on one GPU (gloo): random corpus sizes (down to fewer rows than ranks x K), K, storage and construction path;
indices, scores and embeddings against the unsharded exact order (tests/test_gpu_parallel.py's assertion).
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
        C = int(rng.choice([world, world + 1, 17, 100, 999, 5000, 20000, 65536]))  # (the fixture's scores are distinct up to 65536 rows)
        K = int(rng.integers(1, min(C, 600) + 1))
        try:
            bf16, how = bool(rng.integers(0, 2)), str(rng.choice(["born", "cut"]))
            T.test_sharded_model_forward_topk_bit_exact(world, C, K, int(rng.choice([64, 128])), bf16, how, "gloo")
        except BaseException as e:  # noqa: BLE001
            bad += 1
            print(f"FINDING case {n}: W={world} C={C} K={K} bf16={bf16} {how} | {type(e).__name__} {str(e)[:400]}", flush=True)
        n += 1
    print(f"{n} cases, {bad} findings in {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
