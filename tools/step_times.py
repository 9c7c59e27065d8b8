"""Per-step GPU time distribution of one bench workload (events around every step, no host sync in the loop).
usage: python tools/step_times.py [workload] [steps] [warm-up steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, two_tower_models_amd as A
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cfg = dict(bench.WORKLOADS[wl]); model = bench.build_model(cfg, dev)
opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
batches = bench.make_batches(cfg, 16, dev)
def step(i):
    loss = model.train_forward(*batches[i % 16]); opt.zero_grad(); loss.backward(); opt.step()
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 120
for i in range(warm): step(i)
torch.cuda.synchronize()
evs = []
for i in range(steps):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e); step(i)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
q = lambda f: ms[min(int(f * steps), steps - 1)]
print(f"{wl}: {steps} steps, sweep {opt._sweep_wgs or 768} wgs: min {ms[0]:.3f} p10 {q(.1):.3f} p50 {q(.5):.3f} p90 {q(.9):.3f} max {ms[-1]:.3f} mean {sum(ms)/steps:.3f} ms")
seq = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
print("first 48 in order:", " ".join(f"{v:.2f}" for v in seq[:48]))
