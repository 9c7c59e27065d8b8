"""Randomised check of row-sharded training THROUGH THE MODULE CLASSES at world sizes 2-4 on ONE GPU (ranks as processes
sharing cuda:0, gloo transport -- RCCL refuses two ranks on a device): random table sizes, widths, batch sizes and model
kind (base / hist / debias); every rank's losses, reassembled tables and dense replicas against the oracle's train steps
on the concatenated batch (the assertions of tests/test_gpu_parallel.py, with 0.5 % instead of 0.2 % of a tensor's
elements allowed beyond 5e-6).  Per-rank batches of 8 and more: with a global batch of 2-6 samples most gradient
elements are rounding noise around zero, Adam's first steps turn their SIGN into +-lr, and any two fp32 implementations
-- the CPU port run as 2 ranks included -- then differ in tens of elements per tensor and by 1e-4 in the next loss.
Three or more item features, for a related reason: with one or two, many units of the item MLP's first layer are active
for EVERY sample of the batch, and for those the bias gradient is analytically zero (the in-batch softmax is invariant to
a common shift of the item embeddings) -- noise again (round 5, seed 1: F = 3 still produced two such findings, 3-4 of 256
elements of item_features_arch.0.bias, in 114 cases; from 4 features on none).  History length 2 or more, for the same
reason: with ONE key the attention weights are 1 whatever Q and K are, so the gradients of the Q / K projection weights
are analytically zero -- exactly zero in the oracle's order of operations, rounding noise (1e-12) in the kernels', and
Adam turns the noise into steps of up to lr (one finding: 780 of 49152 in_proj_weight elements off by 1.5e-5, all in the
Q / K rows, which cannot influence any output).  Per-rank batches from 16: seed 7 found the bias class once more at B = 8,
W = 2, F = 4 (3 of 256 elements of item_features_arch.0.bias; 95 cases, nothing else).
    python tools/fuzz_sharded.py [seconds] [seed]"""
import json
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
        kind = str(rng.choice(["base", "base", "hist", "debias"]))
        hist = kind != "base"
        cfg = dict(n_users=int(rng.integers(3, 700)), n_items=int(rng.integers(3, 700)),
                   D=int(rng.choice([32, 64, 128])) if hist else int(rng.choice([8, 24, 40, 64, 128, 160])),
                   F=int(rng.integers(4, 24)), B=int(rng.choice([16, 33, 64, 100])),
                   H=int(rng.choice([2, 4, 9, 50])) if hist else 2)
        if cfg["D"] in (32, 64, 128) and rng.random() < 0.5:
            cfg["mark_from"] = 1  # the owners MARK the rows they serve (optim._SPLIT_MIN_IDS lowered), see tests/test_gpu_parallel.py
        world = int(rng.choice([2, 3, 4]))
        what = f"case {n}: W={world} {kind} {cfg}"
        try:
            T.test_sharded_modules_equal_reference_on_concatenated_batch(world, "json:" + json.dumps([kind, cfg]), "gloo", "torch",
                                                                          bool(rng.integers(0, 2)), outlier_frac=5e-3)
        except BaseException as e:  # noqa: BLE001 -- assertion failures and crashed ranks are both findings
            bad += 1
            print("FINDING", what, "|", type(e).__name__, str(e)[:400], flush=True)
        n += 1
    print(f"{n} cases, {bad} findings in {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
