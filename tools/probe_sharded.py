"""Where does the sharded step's time go at W = 1?  host enqueue time vs GPU time."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
import torch.distributed as dist
from two_tower_models_amd import sharded
dev = torch.device("cuda:0")
dist.init_process_group("nccl", device_id=dev)
cfg = dict(bench.WORKLOADS["P"])
tr = sharded.ShardedTrainer(cfg, dev)
batches = tr.make_batches(8)
for i in range(5):
    tr.step(batches[i % 8])
torch.cuda.synchronize()
for mode in ("async", "sync-each"):
    t0 = time.perf_counter(); host = 0.0
    for i in range(20):
        h0 = time.perf_counter()
        tr.step(batches[i % 8])
        host += time.perf_counter() - h0
        if mode == "sync-each":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{mode}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step, host enqueue {1e3 * host / 20:.3f} ms/step", flush=True)
dist.destroy_process_group()
