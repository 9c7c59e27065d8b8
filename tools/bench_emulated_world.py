"""Per-rank COMPUTE of the row-sharded train step at world size W, on ONE MI355X: torch.distributed
is replaced by a local stand-in whose collectives produce tensors of the right shape without moving
data between processes (all_gather = W copies of the local block, reduce_scatter = the first block,
all_reduce = identity).  The numbers are NOT a training run -- they show what one rank's kernels cost
once the tables are W times thinner and the in-batch negatives W times wider, i.e. the step time at
W GPUs minus the collectives.  Usage: python tools/bench_emulated_world.py [W] [workload]   (TT_ROUTE=alltoall|allgather)"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from two_tower_models_amd import sharded  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
workload = sys.argv[2] if len(sys.argv) > 2 else "P"


class _Done:
    def wait(self):
        return True


def _all_gather(out, x, async_op=False):
    out.view(W, -1).copy_(x.reshape(1, -1).expand(W, -1))
    return _Done() if async_op else None


def _reduce_scatter(out, x, async_op=False):
    out.copy_(x[: out.shape[0]])
    return _Done() if async_op else None


def _all_to_all(out, x, async_op=False):
    # routed lookups.  Float rows: same shape, data irrelevant.  Id lists [W, cap]: a real owner receives from
    # every peer the ~B/W ids of ITS block that the peer looked up; this rank's own chunk for itself has exactly
    # that content and statistics, so every "peer" is given a copy of it.
    if x.dtype == torch.int64 and x.dim() == 1:
        out.view(W, -1).copy_(x.view(W, -1)[0:1].expand(W, -1))
    else:
        out.copy_(x)
    return _Done() if async_op else None


fake = types.SimpleNamespace(all_to_all_single=_all_to_all,
    get_backend=lambda: "emulated", get_world_size=lambda: W, get_rank=lambda: 0, is_initialized=lambda: True,
    all_gather_into_tensor=_all_gather, reduce_scatter_tensor=_reduce_scatter,
    all_reduce=lambda x, op=None, async_op=False: (_Done() if async_op else None),
    broadcast=lambda x, src=0: None, ReduceOp=torch.distributed.ReduceOp, barrier=lambda: None)
sharded.dist = fake

device = torch.device("cuda:0")
cfg = dict(bench.WORKLOADS[workload])
trainer = sharded.ShardedTrainer(cfg, device, negatives="global")
batches = trainer.make_batches(8)
for i in range(5):
    trainer.step(batches[i % 8], batches[(i + 1) % 8])
torch.cuda.synchronize()
steps = 30
t0 = time.perf_counter()
for i in range(steps):
    trainer.step(batches[i % 8], batches[(i + 1) % 8])
host_ms = (time.perf_counter() - t0) / steps * 1e3  # time to ENQUEUE a step (Python + launches)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print(f"  host enqueue time {host_ms:.3f} ms/step")

# where the time goes: events on the main stream around the two logits kernels
marks = []


def _ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append(e)


be = trainer.be
_fwd, _bwd = be.ce_fwd, be.ce_bwd


def ce_fwd(*a, **k):
    _ev()
    out = _fwd(*a, **k)
    _ev()
    return out


def ce_bwd(*a, **k):
    _ev()
    out = _bwd(*a, **k)
    _ev()
    return out


be.ce_fwd, be.ce_bwd = ce_fwd, ce_bwd
acc = [0.0] * 5
for i in range(10):
    marks.clear()
    _ev()
    trainer.step(batches[i % 8], batches[(i + 1) % 8])
    _ev()
    torch.cuda.synchronize()
    for k in range(5):
        acc[k] += marks[k].elapsed_time(marks[k + 1]) / 10
print("  main-stream phases (ms): lookups+towers %.3f | logits fwd + dU %.3f | weights/loss %.3f | logits bwd (dI) %.3f | "
      "towers bwd + row Adam %.3f" % tuple(acc))
print(f"emulated W={W} workload={workload} routing={trainer.routing}: {ms:.3f} ms/step per rank (no collectives) -> "
      f"{cfg['B'] * W / ms * 1e3 / 1e6:.2f} M pairs/s if the collectives were free")
print(f"  bytes this rank would send per step: {trainer.comm_bytes} = {sum(trainer.comm_bytes.values()) / 1e6:.1f} MB")
