"""Randomised differential check of tt_mips_topk against the CPU oracle: random (B, C, D, K, storage), integer-valued
data (exact scores, many ties) so indices AND scores must match bit for bit.   python tools/fuzz_mips.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fixture_gen as fg
import two_tower_models_amd as A
from oracle import cpu_ref as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    D = int(rng.choice([32, 64, 128, 128, 128, 96, 256, 2, 8, 50, 100, 130, 33]))
    B = int(rng.choice([1, 7, 32, 33, 64, 65, 100, 128, 129, 300, 513, 1024, 1100]))
    C = int(rng.integers(200, 60000))
    K = int(rng.integers(1, min(C, 1500)))
    bf16 = bool(rng.integers(0, 2))
    split16 = (not bf16) and D == 128 and bool(rng.integers(0, 2))  # EXPLORATORY: fp32 corpus scored from its fp16 split
    spread = int(rng.choice([3, 5, 9]))
    corpus = torch.from_numpy((fg.hashed_u64((C, D), 100 + n) % np.uint64(spread)).astype(np.float32) - spread // 2)
    q = torch.from_numpy((fg.hashed_u64((B, D), 900 + n) % np.uint64(3)).astype(np.float32) - 1.0)
    m = A.BaselineMIPSModule(corpus_size=C, embedding_dim=D)
    m.corpus = corpus.clone()
    m = m.to("cuda:0")
    if bf16:
        m.use_bf16_storage()
    if split16:
        m.use_split_fp16_scoring()
    idx, sc = m.search(q.to("cuda:0"), K)
    want_idx, want_sc, _ = R.mips_topk(q, corpus, K)
    ok = torch.equal(idx.cpu(), want_idx) and torch.equal(sc.cpu(), want_sc)
    n += 1
    if not ok:
        bad += 1
        print(f"MISMATCH B={B} C={C} D={D} K={K} bf16={bf16} split16={split16} spread={spread} case={n - 1}", flush=True)
print(f"{n} cases, {bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
