"""How far the row-sharded trainer drifts from a float64 restatement of the same steps over a LONG run, with the
fp32-MFMA logits pair and with the exploratory split-fp16 pair (TT_CE_F16X2, csrc/ce_f16x2.hip): if the split pair were a
coarser arithmetic, its loss trajectory and parameters would leave the float64 course sooner than the fp32 pair's do.
Three courses over the same batches and initial values:
    f64   the oracle's train step with every tensor in float64 (the yardstick),
    f32   the oracle's train step in float32 (what the parity tests compare with),
    hip   ShardedTrainer at world size 1 (RCCL group of one), once per logits pair.
Printed per checkpoint: |loss - loss_f64| and the rms / max distance of the two tables from the f64 course.
Test infrastructure (imports the oracle through tests/); not part of the product path.
Usage: python tools/trajectory_f16x2.py [steps=200] [B=1024]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import test_gpu_sharded as T  # noqa: E402
from oracle import cpu_ref as R  # noqa: E402
from test_sharded_cpu import _dense_init  # noqa: E402
from two_tower_models_amd import sharded  # noqa: E402

TABLES = ("user_id_embedding_arch.weight", "item_id_embedding_arch.weight")


def oracle_course(cfg, init, batches, dtype, marks):
    params = {k: v.clone().to(dtype) for k, v in init.items()}
    state = R.AdamState(params)
    w = torch.tensor([0.7], dtype=dtype)
    losses, snaps = [], {}
    for i, b in enumerate(batches):
        b = [t.cpu().to(dtype) if t.is_floating_point() else t.cpu() for t in b]
        losses.append(R.train_step(params, state, b, w))
        if i + 1 in marks:
            snaps[i + 1] = {k: params[k].double().clone() for k in TABLES}
    return losses, snaps


def hip_course(cfg, dense, batches, split, marks):
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    sharded._CE_F16X2 = split
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{T._free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        tr = sharded.ShardedTrainer(cfg, dev, negatives="global", user_value_weights=(0.7,), dense_init=dense)
        init = dict(dense)
        init[TABLES[0]] = tr.users.weight.cpu().clone()
        init[TABLES[1]] = tr.items.weight.cpu().clone()
        losses, snaps, calls = [], {}, []
        real = tr.be.lib.tt_ce16_bwd_recompute
        tr.be.lib.tt_ce16_bwd_recompute = lambda *a: (calls.append(1), real(*a))[1]
        for i, b in enumerate(batches(tr)):
            losses.append(float(tr.step(b)))
            if i + 1 in marks:
                snaps[i + 1] = {TABLES[0]: tr.users.weight.cpu().double(), TABLES[1]: tr.items.weight.cpu().double()}
        tr.be.lib.tt_ce16_bwd_recompute = real
        return init, losses, snaps, len(calls)
    finally:
        dist.destroy_process_group()


def dist_to(snap, ref):
    out = []
    for k in TABLES:
        d = (snap[k][: ref[k].shape[0]] - ref[k]).abs()
        out.append((float(d.pow(2).mean().sqrt()), float(d.max())))
    return out


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    cfg = dict(n_users=3000, n_items=5000, D=128, F=8, B=B, H=2)
    marks = sorted({m for m in (1, 3, 10, 30, 100, 200, 300, 500, 1000) if m <= steps} | {steps})
    dense = _dense_init(cfg)
    mk = lambda tr: tr.make_batches(steps, seed=11)  # noqa: E731
    init, l32h, s32h, _ = hip_course(cfg, dense, mk, False, marks)
    init2, l16h, s16h, used = hip_course(cfg, dense, mk, True, marks)
    assert all(torch.equal(init[k], init2[k]) for k in init), "the two trainers must start from the same values"
    # the batches again, for the oracle (same generator seed and rank)
    class _B:
        pass
    _B.cfg, _B.rank, _B.device = cfg, 0, torch.device("cpu")
    batches = sharded.ShardedTrainer.make_batches(_B, steps, seed=11)
    l64, s64 = oracle_course(cfg, init, batches, torch.float64, marks)
    l32, s32 = oracle_course(cfg, init, batches, torch.float32, marks)
    print(f"cfg {cfg}, {steps} steps, lr 1e-3; split-fp16 backward launches: {used}")
    print(f"{'step':>5} {'loss f64':>12} | {'|dloss| f32 oracle':>18} {'hip fp32 pair':>14} {'hip split pair':>15} | "
          "tables vs f64 course, rms (max):  f32 oracle | hip fp32 pair | hip split pair")
    for m in marks:
        i = m - 1
        row = [dist_to(s[m], s64[m]) for s in (s32, s32h, s16h)]
        cell = lambda r: " ".join(f"{a:.2e} ({b:.1e})" for a, b in r)  # noqa: E731
        print(f"{m:5d} {l64[i]:12.6f} | {abs(l32[i] - l64[i]):18.2e} {abs(l32h[i] - l64[i]):14.2e} {abs(l16h[i] - l64[i]):15.2e} | "
              f"{cell(row[0])} | {cell(row[1])} | {cell(row[2])}")
    a = np.abs(np.array(l32) - np.array(l64))
    b = np.abs(np.array(l32h) - np.array(l64))
    c = np.abs(np.array(l16h) - np.array(l64))
    print(f"max |dloss| over all {steps} steps: f32 oracle {a.max():.2e}, hip fp32 pair {b.max():.2e}, hip split pair {c.max():.2e}; "
          f"mean {a.mean():.2e} / {b.mean():.2e} / {c.mean():.2e}")
