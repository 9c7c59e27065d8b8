"""Randomised check of the whole-step hipGraph (GraphedTrainStep): random model kind / shapes / optimiser schedule,
the same batches through eager steps and through graph replays -- losses and every parameter bit for bit.
    python tools/fuzz_graph.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import two_tower_models_amd as A

DEV = "cuda:0"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    kind = str(rng.choice(["base", "base", "hist", "debias"]))
    D = int(rng.choice([16, 32, 64, 128, 40, 192, 10])) if kind == "base" else int(rng.choice([16, 32, 64, 128]))
    F, B = int(rng.integers(1, 20)), int(rng.choice([1, 16, 64, 100, 256, 700]))
    NU, NI = int(rng.integers(20, 3000)), int(rng.integers(20, 3000))
    H = int(rng.choice([2, 7, 20, 50, 70])) if kind != "base" else 2
    sched = [dict(overlap_sweep=False), dict(overlap_sweep="forward"), dict(lazy=True)][int(rng.integers(0, 3))]
    what = f"case {n}: {kind} D={D} F={F} B={B} NU={NU} NI={NI} H={H} {sched}"
    torch.manual_seed(3000 + n)
    mips = A.BaselineMIPSModule(corpus_size=16, embedding_dim=D)
    kw = dict(num_items=5, user_id_hash_size=NU, user_id_embedding_dim=D, user_features_size=F, item_id_hash_size=NI,
              item_id_embedding_dim=D, item_features_size=F, user_value_weights=[1.0], mips_module=mips)
    make = {"base": lambda: A.TwoTowerBaseRetrieval(**kw), "hist": lambda: A.TwoTowerWithUserHistoryEncoder(user_history_seqlen=H, **kw),
            "debias": lambda: A.TwoTowerWithDebiasing(user_history_seqlen=H, **kw)}[kind]
    proto = make()
    init = {k: v.detach().clone() for k, v in proto.state_dict().items()}
    g = torch.Generator().manual_seed(60 + n)
    batches = [[torch.randint(0, NU, (B,), generator=g), torch.randn(B, F, generator=g), torch.randint(0, NI, (B, H), generator=g),
                torch.randint(0, NI, (B,), generator=g), torch.randn(B, F, generator=g), torch.randint(0, 10, (B,), generator=g),
                torch.randint(0, 2, (B, 1), generator=g).float()] for _ in range(5)]
    batches = [[t.to(DEV) for t in b] for b in batches]
    msgs = []
    try:
        out = []
        for graphed in (False, True):
            model = make()
            model.load_state_dict(init)
            model = model.to(DEV)
            opt = A.DenseExactAdam(model.parameters(), lr=1e-3, **sched)
            losses = []
            if graphed:
                step = A.GraphedTrainStep(model, opt, batches[0], warmup=2)  # 2 real warm-up steps on batch 0
                for b in batches[1:]:
                    losses.append(step(*b).item())
            else:
                for b in [batches[0], batches[0]] + batches[1:]:
                    loss = model.train_forward(*b)
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
                    losses.append(loss.item())
                losses = losses[2:]
            if sched.get("lazy"):
                opt.flush()
            torch.cuda.synchronize()
            out.append((losses, {k: v.cpu() for k, v in model.state_dict().items()}))
            del model, opt
        if out[0][0] != out[1][0]:
            msgs.append(f"losses {out[0][0]} vs {out[1][0]}")
        for k in out[0][1]:
            if not torch.equal(out[0][1][k], out[1][1][k]):
                msgs.append(f"{k}: max diff {float((out[0][1][k] - out[1][1][k]).abs().max()):.2e}")
    except Exception as e:  # noqa: BLE001
        msgs.append(f"{type(e).__name__}: {str(e)[:300]}")
    if msgs:
        bad += 1
        print("FINDING", what, "|", "; ".join(msgs[:5]), flush=True)
    n += 1
print(f"{n} cases, {bad} findings in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
