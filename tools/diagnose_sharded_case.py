"""Diagnosis of a tools/fuzz_sharded.py finding: the same multi-rank case again (gloo ranks on one GPU, HIP backend), and at
every element of ONE tensor that is beyond 5e-6 of the fp32 oracle: the distance of the GPU result and of the fp32 oracle
from the oracle evaluated in float64, next to the element's first-step |gradient| (Adam turns rounding noise on a tiny
gradient into a visible step).  Test infrastructure.
Usage: python tools/diagnose_sharded_case.py WORLD ROUTING 'CFG-JSON' TENSOR-NAME"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import test_gpu_sharded as T
if __name__ == "__main__":
    import torch.multiprocessing as mp
    from oracle import cpu_ref as R
    world, routing, cfg, k = int(sys.argv[1]), sys.argv[2], json.loads(sys.argv[3]), sys.argv[4]
    name = "json:" + json.dumps(cfg)
    outdir = tempfile.mkdtemp()
    mp.spawn(T._multi_worker, args=(world, T._free_port(), outdir, name, "global", "gloo", routing, "torch"), nprocs=world, join=True)
    res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    dense, ut, it = T._multi_init(cfg)
    outs = {}
    grads_first = {}
    for dt in (torch.float32, torch.float64):
        params = {k: v.clone().to(dt) for k, v in dense.items()}
        params["user_id_embedding_arch.weight"] = ut.clone().to(dt)
        params["item_id_embedding_arch.weight"] = it.clone().to(dt)
        state = R.AdamState(params)
        for s in range(T.MULTI_STEPS):
            cat = [torch.cat([res[r]["batches"][s][k] for r in range(world)]) for k in range(7)]
            cat = [t.to(dt) if t.is_floating_point() else t for t in cat]
            if s == 0 and dt == torch.float64:
                leaves = {k: v.detach().requires_grad_(True) for k, v in params.items()}
                loss = R.train_forward(leaves, cat, torch.tensor([0.7], dtype=dt))
                g = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
                grads_first = {k: gi for k, gi in zip(leaves, g)}
            R.train_step(params, state, cat, torch.tensor([0.7], dtype=dt))
        outs[dt] = params
    got = res[0]["dense"][k].double()
    o32, o64 = outs[torch.float32][k].double(), outs[torch.float64][k]
    err = (got - o32).abs()
    idx = (err > 5e-6).nonzero()
    g0 = grads_first[k].abs()
    print(f"{k}: {len(idx)} elements of {err.numel()} beyond 5e-6 vs the fp32 oracle; median |g| of the first step over the tensor {float(g0.median()):.3e}")
    for i in idx:
        i = tuple(int(x) for x in i)
        print(f"  {i}: |GPU - f32 oracle| {float(err[i]):.2e}   |GPU - f64| {float((got[i] - o64[i]).abs()):.2e}   |f32 oracle - f64| {float((o32[i] - o64[i]).abs()):.2e}   first-step |g| {float(g0[i]):.2e}")
    print(f"over the whole tensor: max |GPU - f64| {float((got - o64).abs().max()):.2e}, max |f32 oracle - f64| {float((o32 - o64).abs().max()):.2e}; "
          f"elements beyond 5e-6 of f64: GPU {int(((got - o64).abs() > 5e-6).sum())}, f32 oracle {int(((o32 - o64).abs() > 5e-6).sum())}")
