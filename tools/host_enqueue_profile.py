import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, two_tower_models_amd as A
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
for wl in ("C2", "P"):
    cfg = dict(bench.WORKLOADS[wl]); model = bench.build_model(cfg, dev)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward")
    batches = bench.make_batches(cfg, 16, dev); total = torch.zeros((), device=dev)
    def step(i):
        loss = model.train_forward(*batches[i % 16]); opt.zero_grad(); loss.backward(); opt.step(); total.add_(loss.detach())
    for i in range(10): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): step(i)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{wl}: host enqueue {1e3*(t1-t0)/100:.3f} ms/step, total {1e3*(t2-t0)/100:.3f} ms/step")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(50): step(i)
    pr.disable(); torch.cuda.synchronize()
    if wl == "C2": pstats.Stats(pr).sort_stats("tottime").print_stats(18)
    del model, opt; torch.cuda.empty_cache()
