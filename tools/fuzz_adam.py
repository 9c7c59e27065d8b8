"""Randomised multi-step check of the optimiser schedules: random model kind / shapes, 2-5 train steps whose batches
share some ids and not others (rows that are looked up, rest for a step or two on momentum alone, and are looked up
again), random DenseExactAdam schedule (serial sweep, sweep started by zero_grad, sweep started by the forward,
deferred = lazy + flush).  Against the oracle's torch.optim.Adam-semantics trajectory: every loss (1e-4), never
looked-up rows bit-identical, every other table row and every dense parameter within 5e-6 for all but 0.5 % of a
tensor's elements and within steps * 2.2 lr for all (Adam's early steps are sign-sensitive on near-zero gradients,
tests/test_gpu_models.py).  The schedules must also agree with EACH OTHER bit for bit.
    python tools/fuzz_adam.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import two_tower_models_amd as A
from oracle import cpu_ref as R

DEV = "cuda:0"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
# "marked": the forward schedule with the big-lookup threshold lowered to 1 -- rows marked for the sweep to step over where the
# row width allows (32 / 64 / 128), moments parked on the sweep's stream otherwise
SCHEDULES = [dict(overlap_sweep=False), dict(overlap_sweep=True), dict(overlap_sweep="forward"), dict(lazy=True),
             dict(overlap_sweep="forward", marked=True)]
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    kind = str(rng.choice(["base", "base", "hist"]))
    D = int(rng.choice([16, 32, 64, 128, 40])) if kind == "base" else int(rng.choice([16, 32, 64, 128]))
    F, B = int(rng.integers(3, 20)), int(rng.choice([16, 64, 100, 256, 700]))
    NU, NI = int(rng.integers(20, 3000)), int(rng.integers(20, 3000))
    H = int(rng.choice([2, 7, 20])) if kind == "hist" else 2
    steps = int(rng.integers(2, 6))
    what = f"case {n}: {kind} D={D} F={F} B={B} NU={NU} NI={NI} H={H} steps={steps}"
    torch.manual_seed(2000 + n)
    mips = A.BaselineMIPSModule(corpus_size=16, embedding_dim=D)
    kw = dict(num_items=5, user_id_hash_size=NU, user_id_embedding_dim=D, user_features_size=F, item_id_hash_size=NI,
              item_id_embedding_dim=D, item_features_size=F, user_value_weights=[1.0], mips_module=mips)
    make = (lambda: A.TwoTowerBaseRetrieval(**kw)) if kind == "base" else (lambda: A.TwoTowerWithUserHistoryEncoder(user_history_seqlen=H, **kw))
    proto = make()
    with torch.no_grad():
        for name, p in proto.named_parameters():
            if name.endswith("tower_arch.weight") or "embedding_arch" in name:
                p.mul_(0.3)
    init = {k: v.detach().clone() for k, v in proto.state_dict().items()}
    g = torch.Generator().manual_seed(50 + n)
    batches = []
    for s in range(steps):
        b = [torch.randint(0, NU, (B,), generator=g), torch.randn(B, F, generator=g), torch.randint(0, NI, (B, H), generator=g),
             torch.randint(0, NI, (B,), generator=g), torch.randn(B, F, generator=g), torch.randint(0, 10, (B,), generator=g),
             torch.randint(0, 2, (B, 1), generator=g).float()]
        if s and rng.random() < 0.7:  # ids of an earlier step come back
            src = batches[int(rng.integers(0, s))]
            k = int(rng.integers(1, B + 1))
            b[0][:k], b[3][:k] = src[0][:k], src[3][:k]
        batches.append(b)
    fkw = dict(with_history=True, heads=4, pos_table=R.positional_table(H, D)) if kind == "hist" else {}
    params = {k: v.clone() for k, v in init.items()}
    state = R.AdamState(params)
    want = [R.train_step(params, state, b, torch.tensor([1.0]), **fkw) for b in batches]
    results, msgs = [], []
    try:
        for sched in SCHEDULES:
            model = make()
            model.load_state_dict(init)
            model = model.to(DEV)
            A.optim._SPLIT_MIN_IDS = 1 if sched.get("marked") else 65536
            opt = A.DenseExactAdam(model.parameters(), lr=1e-3, **{k: v for k, v in sched.items() if k != "marked"})
            losses = []
            for b in batches:
                loss = model.train_forward(*[t.to(DEV) for t in b])
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses.append(loss.item())
            if sched.get("lazy"):
                opt.flush()
            torch.cuda.synchronize()
            sd = {k: v.cpu() for k, v in model.state_dict().items()}
            results.append(sd)
            tag = str(sched)
            if not np.allclose(losses, want, atol=1e-4):
                msgs.append(f"{tag}: losses {losses} vs {want}")
            touched = {"user_id_embedding_arch.weight": torch.unique(torch.cat([b[0] for b in batches])),
                       "item_id_embedding_arch.weight": torch.unique(torch.cat([torch.cat([b[3], b[2].flatten()]) if kind == "hist" else b[3] for b in batches]))}
            for k, v in sd.items():
                if k in touched:
                    mask = torch.ones(v.shape[0], dtype=torch.bool)
                    mask[touched[k]] = False
                    if not torch.equal(v[mask], init[k][mask]):
                        msgs.append(f"{tag}: {k}: a never-looked-up row changed")
                err = (v - params[k]).abs()
                noise_only = k in ("item_tower_arch.bias", "item_features_arch.2.bias") or k.endswith("in_proj_bias")
                if float(err.max()) > 2.2e-3 * steps:
                    msgs.append(f"{tag}: {k}: max err {float(err.max()):.2e}")
                elif k == "item_features_arch.0.bias" and int((err > 5e-6).sum()) <= (8 if B > 16 else 16):
                    # units active for every item of a batch: analytically zero bias gradient.  With 16
                    # items and 3 features a dozen of the 256 units can be (seeds 611 / 612: 10 and 9 of them, every schedule
                    # alike, the serial one included)
                    pass
                elif not noise_only and int((err > 5e-6).sum()) > max(2, int(5e-3 * err.numel())):
                    msgs.append(f"{tag}: {k}: {int((err > 5e-6).sum())} of {err.numel()} beyond 5e-6")
        for i in range(1, len(results)):  # the schedules are the same arithmetic in a different order of launches
            for k in results[0]:
                if not torch.equal(results[0][k], results[i][k]):
                    msgs.append(f"{SCHEDULES[i]} differs from {SCHEDULES[0]} in {k}: max {float((results[0][k] - results[i][k]).abs().max()):.2e}")
    except Exception as e:  # noqa: BLE001
        msgs.append(f"{type(e).__name__}: {str(e)[:300]}")
    if msgs:
        bad += 1
        print("FINDING", what, "|", "; ".join(msgs[:6]), flush=True)
    n += 1
print(f"{n} cases, {bad} findings in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
