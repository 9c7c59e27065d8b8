#!/usr/bin/env python
"""bench.py -- user-item pairs/s through one in-batch-softmax TRAIN STEP (forward,
zero_grad, backward, dense-exact Adam: the body of ref:train/train.py:112-125) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric: B=8192, d=128, 10 M-row item table): TwoTowerBaseRetrieval,
N_u = 1 M users, N_i = 10 M items, D = 128, F = 8, T = 1, B = 8192 per GPU, synthetic
batches of ref:train/train.py:47-65's distributions pre-generated in HBM, random-init
weights.  With N > 1 the tables are row-sharded (N_i/N rows per GPU), each rank feeds its
own B = 8192 batch and the in-batch negatives are global (N*B items per user): weak scaling
in the batch, whole-job pairs/s = N*B*steps / time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = the Adam table sweep, HBM
bound, timed live with HIP events on its own stream) and `cpu_baseline` (the CPU oracle
oracle/cpu_ref.py -- kind "port" -- timed on this box's host cores, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle errors
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32 MFMA / vector dense peak
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured float4 copy)

WORKLOADS = {
    # name: (n_users, n_items, D, F, B, H, model)
    "P": dict(n_users=1_000_000, n_items=10_000_000, D=128, F=8, B=8192, H=4, model="base"),
    "C2": dict(n_users=1_000_000, n_items=1_000_000, D=128, F=8, B=4096, H=4, model="base"),
    "C3": dict(n_users=1_000_000, n_items=1_000_000, D=128, F=8, B=4096, H=50, model="hist"),
    # BASELINE config 4: 100 M items (row-sharded x8 in the reference plan; 155 GB of p, m, v also fits ONE
    # MI355X's 288 GB, which is what `--workload C4 --gpus 1` runs)
    "C4": dict(n_users=1_000_000, n_items=100_000_000, D=128, F=8, B=8192, H=4, model="base"),
    # logits-bound shape (what every rank sees once the tables are sharded thin): small tables, full batch
    # BASELINE config 5's model in TRAINING (its loss head is SURVEY 8f item 2): C3 plus the debias head
    "C5T": dict(n_users=1_000_000, n_items=1_000_000, D=128, F=8, B=4096, H=50, model="debias"),
    "CE": dict(n_users=100_000, n_items=100_000, D=128, F=8, B=8192, H=4, model="base"),
    # BASELINE config 5 as it is written: TwoTowerWithDebiasing.forward() -> baseline_mips_module top-K = 1000 over a 10 M-row
    # bf16 corpus; with --gpus N the tables AND the corpus are row-sharded (C / N rows per GPU), every rank brings its own
    # 1024 users.  An INFERENCE workload: metric = queries/s, a "step" = one model.forward() per rank
    "C5": dict(n_users=1_000_000, n_items=10_000_000, D=128, F=8, B=1024, H=50, model="debias",
               mips=dict(C=10_000_000, K=1000, bf16=True)),
    "tiny": dict(n_users=1024, n_items=10_000, D=32, F=8, B=128, H=4, model="base"),
    # (tests: the C5 line's contract at a size the 1-GPU box runs as two gloo ranks in seconds)
    "C5tiny": dict(n_users=1024, n_items=10_000, D=128, F=8, B=64, H=6, model="debias", mips=dict(C=20_000, K=100, bf16=True)),
}


def make_batches(cfg, n, device, seed=1234):
    """Distributions of ref:train/train.py:47-65; labels as [B,1] (real weighting path)."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    B, F, H = cfg["B"], cfg["F"], cfg["H"]
    out = []
    for _ in range(n):
        b = (
            torch.randint(0, cfg["n_users"], (B,), generator=gen),
            torch.randn(B, F, generator=gen),
            torch.randint(0, cfg["n_items"], (B, H), generator=gen),
            torch.randint(0, cfg["n_items"], (B,), generator=gen),
            torch.randn(B, F, generator=gen),
            torch.randint(0, 10, (B,), generator=gen),
            torch.randint(0, 2, (B, 1), generator=gen).float(),
        )
        out.append(tuple(t.to(device) for t in b))
    return out


def build_model(cfg, device, seed=0):
    import two_tower_models_amd as A
    torch.manual_seed(seed)
    mp = cfg.get("mips") or dict(C=1024, K=10, bf16=False)
    with torch.device(device):  # initialise the 5.6 GB of tables directly in HBM
        mips = A.BaselineMIPSModule(corpus_size=mp["C"], embedding_dim=cfg["D"])
        if mp["bf16"]:
            mips.use_bf16_storage()
        kw = dict(num_items=mp["K"], user_id_hash_size=cfg["n_users"], user_id_embedding_dim=cfg["D"],
                  user_features_size=cfg["F"], item_id_hash_size=cfg["n_items"],
                  item_id_embedding_dim=cfg["D"], item_features_size=cfg["F"],
                  user_value_weights=[1.0], mips_module=mips)
        if cfg["model"] == "hist":
            model = A.TwoTowerWithUserHistoryEncoder(user_history_seqlen=cfg["H"], **kw)
        elif cfg["model"] == "debias":
            model = A.TwoTowerWithDebiasing(user_history_seqlen=cfg["H"], **kw)
        else:
            model = A.TwoTowerBaseRetrieval(**kw)
    return model.to(device)


def pmc_traffic_bytes(path, workload, world):
    """roofline.traffic: the committed PMC measurement of the sweep's HBM bytes per launch (profiles/pmc_traffic.json,
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command).  The file is keyed by "<workload>_gpus<N>"; a
    flat record (round 3's refresh wrote one, and the lookup then silently returned null) counts as "P_gpus1"."""
    if not os.path.exists(path):
        return None
    data = json.load(open(path))
    rec = data.get(f"{workload}_gpus{world}")
    if rec is None and "hbm_bytes_per_launch" in data and (workload, world) == ("P", 1):
        rec = data
    return rec.get("hbm_bytes_per_launch") if rec else None


def sysfs_clocks():
    """Current sclk / mclk (MHz) and board power of card 0 from the amdgpu sysfs files ('*' marks the active DPM level);
    {} where the files are not readable.  Read before and after the timed region: a box that streams slower at the same
    clocks is a slower box, one that dropped its memory clock is a throttling one."""
    import glob
    out = {}
    for key, name in (("sclk_MHz", "pp_dpm_sclk"), ("mclk_MHz", "pp_dpm_mclk")):
        for path in sorted(glob.glob(f"/sys/class/drm/card*/device/{name}")):
            try:
                for line in open(path):
                    if "*" in line:
                        out[key] = int(line.split(":")[1].strip().split("M")[0])
                break
            except (OSError, ValueError, IndexError):
                continue
    # (the DPM level marked '*' is an instantaneous sample and reads the idle level more often than not, even with work queued;
    # hwmon's freq inputs -- where the driver exposes them -- are the clocks actually running)
    for key, name in (("sclk_hwmon_MHz", "freq1_input"), ("mclk_hwmon_MHz", "freq2_input")):
        for path in sorted(glob.glob(f"/sys/class/drm/card*/device/hwmon/hwmon*/{name}")):
            try:
                out[key] = round(int(open(path).read()) / 1e6)
                break
            except (OSError, ValueError):
                continue
    for path in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")):
        try:
            out["power_W"] = round(int(open(path).read()) / 1e6, 1)
            break
        except (OSError, ValueError):
            continue
    return out


def hbm_copy_calibration(lib, N, device, nbytes, launches=7):
    """The box's own HBM streaming rate, measured in THIS process right after the timed steps: tt_stream_copy (16-byte
    non-temporal loads / stores, one-shot grid: the fastest of the copies tools/copy_probe.hip tries) of a buffer as large as
    the sweep's footprint -- `nbytes` / 2 read and
    `nbytes` / 2 written per launch.  Median of `launches` HIP-event pairs.  The guide quotes 6.29 TB/s for this
    measurement; the sweep's `frac_of_copy` says how the kernel does against what this box can stream at all."""
    half = int(nbytes // 2) // 16 * 16
    src = torch.empty(half, dtype=torch.uint8, device=device)
    dst = torch.empty(half, dtype=torch.uint8, device=device)
    src.zero_()
    dst.zero_()
    evs = []
    for _ in range(launches + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        N.check(lib.tt_stream_copy(src.data_ptr(), dst.data_ptr(), half, N.stream()), "tt_stream_copy")
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs[1:])
    del src, dst
    return {"GBps": round(2.0 * half / (ms[len(ms) // 2] * 1e-3) / 1e9, 1), "best_GBps": round(2.0 * half / (ms[0] * 1e-3) / 1e9, 1),
            "bytes_per_launch": 2 * half, "launches": launches, "median_launch_ms": round(ms[len(ms) // 2], 4)}


def mfma_sustained_peak(lib, N, device, dtype, iters=4000):
    """TFLOP/s of the register-only MFMA loop (tt_mfma_probe: random operands, no LDS, no HBM, no epilogue) in THIS process:
    what the matrix pipe sustains at this box's power budget -- the ceiling a kernel priced against the spec peak has."""
    sink = torch.empty(512 * 256, dtype=torch.float32, device=device)
    N.check(lib.tt_mfma_probe(dtype, 200, sink.data_ptr(), sink.numel(), N.stream()), "tt_mfma_probe")
    best = 0.0
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        N.check(lib.tt_mfma_probe(dtype, iters, sink.data_ptr(), sink.numel(), N.stream()), "tt_mfma_probe")
        b.record()
        b.synchronize()
        best = max(best, lib.tt_mfma_probe_flops(dtype, iters) / (a.elapsed_time(b) * 1e-3) / 1e12)
    return round(best, 1)


def rccl_identity(device, world):
    """What the collective library reports about the group this process is in (so that "RCCL saw N ranks" can be read off
    the line): the sum of a ones-vector all-reduced over the RCCL process group (= the number of ranks that took part in an
    RCCL collective), ncclCommCount / ncclCommUserRank of the C ABI's own communicator when that is the transport
    (`--transport native`; no second communicator is created just to ask), and every rank's device ordinal and PCI bus id."""
    import ctypes
    import torch.distributed as dist
    from two_tower_models_amd import collectives
    out = {"backend": dist.get_backend(), "process_group_world_size": dist.get_world_size()}
    ones = torch.ones(1, device=device)
    dist.all_reduce(ones)
    out["ranks_in_an_allreduce_of_ones"] = int(ones.item())
    try:
        out["nccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        pass
    if collectives._NATIVE is not None:
        out["ncclCommUserRank"], out["ncclCommCount"] = collectives._NATIVE.size()
    else:
        out["ncclCommCount"] = None  # (reported by the tt_comm_* communicator: run with --transport native)
    bus = "unknown"
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device.index or 0)) == 0:
            bus = buf.value.decode()
    except OSError:
        pass
    mine = f"rank {dist.get_rank()}: cuda:{device.index} pci {bus} pid {os.getpid()}"
    every = [None] * world
    dist.all_gather_object(every, mine)
    out["ranks"] = every
    out["distinct_pci_bus_ids"] = len({e.split(" pci ")[1].split(" pid")[0] for e in every})
    return out


def algorithmic_sweep_bytes(cfg, world):
    """Adam reads+writes p, m, v of EVERY table row: 24 B per element."""
    return 24.0 * (cfg["n_users"] + cfg["n_items"]) * cfg["D"] / world


def cpu_baseline(cfg, seconds_budget=25.0):
    """The CPU oracle (port of the reference's path) on this box's host cores, same shapes,
    bounded sample.  Tables are shrunk only if host RAM cannot hold p, m, v and the gradient."""
    import psutil
    from oracle import cpu_ref as R
    cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    torch.set_num_threads(cores)
    n_items, n_users, D, F, B = cfg["n_items"], cfg["n_users"], cfg["D"], cfg["F"], cfg["B"]
    need = lambda ni: 4 * 4 * (ni + n_users) * D * 1.3
    avail = psutil.virtual_memory().available
    while need(n_items) > 0.5 * avail and n_items > 100_000:
        n_items //= 2
    torch.manual_seed(0)
    p = {
        "user_id_embedding_arch.weight": torch.randn(n_users, D),
        "item_id_embedding_arch.weight": torch.randn(n_items, D),
    }
    for side, fin in (("user", F), ("item", F)):
        p[f"{side}_features_arch.0.weight"] = torch.randn(256, fin) * 0.3
        p[f"{side}_features_arch.0.bias"] = torch.zeros(256)
        p[f"{side}_features_arch.2.weight"] = torch.randn(D, 256) * 0.06
        p[f"{side}_features_arch.2.bias"] = torch.zeros(D)
        p[f"{side}_tower_arch.weight"] = torch.randn(D, 2 * D) * 0.06
        p[f"{side}_tower_arch.bias"] = torch.zeros(D)
    state = R.AdamState(p)
    small = dict(cfg, n_items=n_items)
    batches = make_batches(small, 2, "cpu")
    uvw = torch.tensor([1.0])
    R.train_step(p, state, batches[0], uvw)  # warm-up (page-faults the 4 big arrays in)
    # thread sweep: a streaming optimiser step on a 2-socket box is not fastest with every core (128
    # untuned threads were slower than the reference on 8 vCPUs); one step each at cores, cores/2, cores/4 (>= 8),
    # the rest of the budget at the best count
    sweep = {}
    for n in sorted({cores, max(cores // 2, 8), max(cores // 4, 8)}, reverse=True):
        if n > cores:
            continue
        torch.set_num_threads(n)
        t1 = time.perf_counter()
        R.train_step(p, state, batches[len(sweep) % 2], uvw)
        sweep[n] = time.perf_counter() - t1
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    steps = 0
    while True:
        R.train_step(p, state, batches[steps % 2], uvw)
        steps += 1
        if time.perf_counter() - t0 > seconds_budget * 0.4 or steps >= 8:
            break
    dt = time.perf_counter() - t0
    if sweep[best] < dt / steps:  # the sweep's own step was the fastest sample
        dt, steps = sweep[best], 1
    cores_used = best
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {
        "value": B * steps / dt, "unit": "pairs/s", "cores": cores_used, "kind": "port", "cpu_model": cpu_model,
        "cpu_steps": steps, "physical_cores": cores,
        "thread_sweep_ms_per_step": {str(k): round(v * 1e3) for k, v in sweep.items()},
        "sample": f"{steps} train steps of oracle/cpu_ref.py (torch CPU, best of a {sorted(sweep)}-thread sweep: {cores_used} threads), B={B}, D={D}, "
                  f"N_u={n_users}, N_i={n_items}" + ("" if n_items == cfg["n_items"] else " (shrunk to fit host RAM)")
                  + f", {dt / steps * 1e3:.0f} ms/step",
    }


MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak


def cpu_baseline_small(name, n_steps=1):
    """`cpu_baseline` beside a SECONDARY train workload ("same shapes ... in the same run"): the CPU oracle's
    train step -- with the history encoder where the workload has one -- on this box's host cores, one warm-up step +
    `n_steps` timed, thread count stated.  The parameters come from the package's own module built on the CPU (same
    shapes and init as the GPU run; only its state_dict is used, the HIP path never runs on CPU tensors)."""
    import psutil
    import two_tower_models_amd as A
    from oracle import cpu_ref as R
    cfg = dict(WORKLOADS[name])
    cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    threads = max(min(cores // 4, 32), min(cores, 8))  # the P baseline's thread sweep settles at cores / 4 on this class of box
    torch.set_num_threads(threads)
    model = build_model(cfg, "cpu")
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    hist = cfg["model"] != "base"
    kw = dict(with_history=True, heads=4, pos_table=R.positional_table(cfg["H"], cfg["D"])) if hist else {}
    state = R.AdamState(params)
    batches = make_batches(cfg, 2, "cpu")
    uvw = torch.tensor([1.0])
    R.train_step(params, state, batches[0], uvw, **kw)
    t0 = time.perf_counter()
    for i in range(n_steps):
        R.train_step(params, state, batches[(i + 1) % 2], uvw, **kw)
    dt = (time.perf_counter() - t0) / n_steps
    return {"value": round(cfg["B"] / dt, 1), "unit": "pairs/s", "cores": threads, "kind": "port", "physical_cores": cores,
            "sample": f"{n_steps} train step(s) of oracle/cpu_ref.py after one warm-up, {threads} threads, workload {name}"
                      f" (B={cfg['B']}, N_u={cfg['n_users']}, N_i={cfg['n_items']}, D={cfg['D']}"
                      + (f", H={cfg['H']}, 3 attention layers x 4 heads" if hist else "") + f"), {dt * 1e3:.0f} ms/step"}


def cpu_baseline_mips(corpus_cpu, K, n_queries=64):
    """`cpu_baseline` beside the MIPS secondary: oracle/cpu_ref.mips_topk (= torch.topk(q @ corpus.T), the reference's
    ref:src/baseline_mips_module.py:57-61, in query chunks -- the reference itself materialises [B, C]) on `n_queries`
    queries over the SAME fp32 corpus, host cores."""
    import psutil
    from oracle import cpu_ref as R
    cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    threads = max(min(cores // 2, 64), min(cores, 8))
    torch.set_num_threads(threads)
    q = torch.randn(n_queries, corpus_cpu.shape[1], generator=torch.Generator().manual_seed(1))
    R.mips_topk(q[:8], corpus_cpu, K, chunk=8)
    t0 = time.perf_counter()
    R.mips_topk(q, corpus_cpu, K, chunk=16)
    dt = time.perf_counter() - t0
    return {"value": round(n_queries / dt, 1), "unit": "queries/s", "cores": threads, "kind": "port", "physical_cores": cores,
            "sample": f"{n_queries} queries, C={corpus_cpu.shape[0]}, D={corpus_cpu.shape[1]}, K={K}, fp32, chunks of 16 queries, "
                      f"{threads} threads, {dt:.1f} s"}


def build_sharded(cfg, device, rank):
    """The workload's model as THIS rank's member of a row-sharded group (parallel.row_sharded: each table's block is
    born on its owner, never whole) + its optimiser; replicated parameters are rank 0's."""
    import two_tower_models_amd as A
    from two_tower_models_amd import parallel
    with parallel.row_sharded():
        model = build_model(cfg, device, seed=1000 + rank)  # (different rows on every rank; dense: broadcast below)
    parallel.shard_model_(model)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3)  # row-sharded tables: the forward-announced dense-exact schedule
    return model, opt


def sharded_step_fn(model, opt, total_loss, watchdog=None):
    """The reference loop body (ref:train/train.py:112-125) + the announcement of the NEXT batch's lookups."""
    from two_tower_models_amd import parallel

    def step(batch, nxt=None):
        loss = model.train_forward(*batch)
        if nxt is not None:  # the next batch's routes are planned underneath this step (no host wait)
            parallel.plan_ahead(model._lookup_plan(nxt[0], nxt[2], nxt[3]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        total_loss.add_(loss.detach())
        if watchdog is not None:
            watchdog.mark()

    return step


def sharded_check(device, rank, world, steps=3):
    """`--check`: the first contact of a new node with the sharded path, made boring.  Every rank builds the SAME small
    model (seeded), keeps its row blocks, and runs `steps` train steps on its own batch; rank 0 then repeats the steps on
    the CONCATENATED batches through the single-process module path (the path tests/ pin to the oracle).  The W-rank losses
    must equal the 1-rank losses to 1e-4 (the north star's loss tolerance)."""
    import two_tower_models_amd as A
    from two_tower_models_amd import parallel
    cfg = dict(n_users=4096, n_items=8192, D=128, F=8, B=256, H=4, model="base")

    def whole():
        model = build_model(cfg, "cpu", seed=7)
        with torch.no_grad():  # logits O(1): differences show up in the loss, not in saturation
            for n, p in model.named_parameters():
                p.mul_(0.5 if n.endswith("embedding_arch.weight") else 0.5 if n.endswith("tower_arch.weight") else 1.0)
        return model.to(device)

    per_rank = [make_batches(cfg, steps, "cpu", seed=77 + 1000 * r) for r in range(world)]
    model = parallel.shard_model_(whole())
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3)
    got = []
    for s_ in range(steps):
        loss = model.train_forward(*[t.to(device) for t in per_rank[rank][s_]])
        opt.zero_grad()
        loss.backward()
        opt.step()
        got.append(float(loss))
    del model, opt
    out = {"ok": True, "steps": steps, "shape": cfg, "sharded_losses": got}
    if rank == 0:
        single = whole()
        opt1 = A.DenseExactAdam(single.parameters(), lr=1e-3)
        want = []
        for s_ in range(steps):
            cat = [torch.cat([per_rank[r][s_][k] for r in range(world)]).to(device) for k in range(7)]
            loss = single.train_forward(*cat)
            opt1.zero_grad()
            loss.backward()
            opt1.step()
            want.append(float(loss))
        err = max(abs(a - b) for a, b in zip(got, want))
        out.update(single_process_losses=want, max_abs_diff=err, ok=bool(err < 1e-4))
    flag = torch.tensor([1.0 if out["ok"] else 0.0], device=device)
    parallel.C.broadcast_(flag, src=0)
    out["ok"] = bool(flag.item() > 0.5)
    return out


def _timed_sharded_w1(device, steps=20, warmup=5):
    """The row-sharded module path at W = 1 over RCCL (`bench.py --sharded --gpus 1`) at the P shapes: the N = 1 point the
    scaling curve starts from must equal the single-GPU headline (tests/test_gpu_bench_contract.py holds it to 3 %)."""
    import torch.distributed as dist
    if dist.is_initialized():
        raise RuntimeError("a process group is already up")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.update(RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=device)
    try:
        cfg = dict(WORKLOADS["P"])
        model, opt = build_sharded(cfg, device, 0)
        batches = make_batches(cfg, 16, device)
        total = torch.zeros((), device=device)
        step = sharded_step_fn(model, opt, total)
        for i in range(warmup):
            step(batches[i % 16], batches[(i + 1) % 16])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(batches[(warmup + i) % 16], batches[(warmup + i + 1) % 16])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = {"workload": "P through the row-sharded MODULE path (parallel.row_sharded + TwoTowerBaseRetrieval + DenseExactAdam), "
                           "world size 1, RCCL process group (the N = 1 point of the scaling curve)",
               "pairs_per_s": round(cfg["B"] * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
               "warmup": warmup}
        del model, opt, batches, step
    finally:
        dist.destroy_process_group()
    return out


def _expected_idle(t, r):
    """Mean idle time of a row that is looked up at step t and was looked up before, ids uniform with mean recurrence r."""
    x = t / r
    return r * (1.0 - x * math.exp(-x) / (1.0 - math.exp(-x)))


def _tower_flops_fwd(B, D, F, tower_in):
    return 2.0 * B * (F * 256 + 256 * D) + 2.0 * B * tower_in * D


def step_flops(cfg):
    """Dense-contraction flops of one train step, two ways:
      executed -- what this build's kernels actually multiply after its algebraic work removal: towers x3 (fwd, data
                  grad, weight grad), FOUR logit-sized products (fwd S and E = P.I; bwd S again and dI), and for the
                  history model 2 full attention layers (the last layer is consumed at row 0 only and runs collapsed;
                  out-projections are folded into the next in-projection) with an attention backward of 5 H^2 products;
      survey   -- SURVEY.md's count: 3 x the reference's forward contractions (+ nothing recomputed)."""
    B, D, F, H = cfg["B"], cfg["D"], cfg["F"], cfg["H"]
    hist = cfg["model"] != "base"
    logits = 2.0 * B * B * D
    towers = _tower_flops_fwd(B, D, F, 4 * D if hist else 2 * D) + _tower_flops_fwd(B, D, F, 2 * D)
    executed, survey = 3.0 * towers + 4.0 * logits, 3.0 * (towers + logits)
    if hist:
        L = 3
        inproj, attn, outproj = 2.0 * B * H * D * 3 * D, 4.0 * B * H * H * D, 2.0 * B * H * D * D
        last_fwd = 4.0 * B * H * D * 4 + 8.0 * B * D * D
        executed += (L - 1) * (inproj + attn) + last_fwd + (L - 1) * (2.0 * inproj + 2.5 * attn) + 2.0 * last_fwd
        survey += 3.0 * L * (inproj + attn + outproj)
    return executed, survey


def _timed_train(name, device, steps, warmup, lazy=False, fresh_ids=False, steady=False, graph=False):
    """One module-path train workload, timed like the headline (batches resident, K steps between syncs).
    `fresh_ids`: every step looks up NEW uniform ids (generated on the device outside nothing -- inside the timed
    region, 3 randint launches per step) instead of cycling 8 batches: the steady state of the deferred schedule,
    where a looked-up row has idled ~N / B steps."""
    import two_tower_models_amd as A
    cfg = dict(WORKLOADS[name])
    model = build_model(cfg, device)
    opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep="forward", lazy=lazy)
    batches = make_batches(cfg, 8, device)
    total = torch.zeros((), device=device)
    id_gen = torch.Generator(device=device).manual_seed(4321)
    fresh = {}

    def batch_at(i):
        if not fresh_ids:
            return batches[i % len(batches)]
        if i not in fresh:
            b = list(batches[i % len(batches)])
            b[0] = torch.randint(0, cfg["n_users"], tuple(b[0].shape), device=device, generator=id_gen)
            b[3] = torch.randint(0, cfg["n_items"], tuple(b[3].shape), device=device, generator=id_gen)
            fresh.pop(i - 2, None)
            fresh[i] = b
        return fresh[i]

    def step(i):
        b, nxt = batch_at(i), batch_at(i + 1)
        loss = model.train_forward(*b)
        if lazy:
            opt.prefetch_rows(model._lookup_plan(nxt[0], nxt[2], nxt[3]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        total.add_(loss.detach())

    if graph:  # the whole step as ONE hipGraph launch (graphs.GraphedTrainStep: the ~90 launches of a step become one)
        graphed = A.GraphedTrainStep(model, opt, batches[0], warmup=3)

        def step(i):  # noqa: F811
            total.add_(graphed(*batch_at(i)))

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    import ctypes as C
    from two_tower_models_amd import _native as N
    lib = N.load()
    # no library profiling events here: an event pair around every attention / projection kernel of a 64-kernel step costs
    # it 3 - 5 % (3.49 vs 3.35 ms: the events end the overlap between consecutive kernels)
    lib.tt_profile_enable(0)
    if not lazy:
        opt.keep_sweep_events(True)  # the sweep's launch duration from the optimiser's own event pair: no extra events
    import gc
    gc.collect()
    gc.disable()  # a cyclic-GC pause inside a 25 ms timed window of a host-bound loop is a 30 % error (seen: 1.38 vs 1.75 ms)
    flush_s = None
    try:
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        if lazy and not steady:
            opt.flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if lazy and steady:  # the flush is paid once per RUN (before anything reads a table as a whole), not per step
            t1 = time.perf_counter()
            opt.flush()
            torch.cuda.synchronize()
            flush_s = time.perf_counter() - t1
    finally:
        gc.enable()
    sw_ms, sw_cnt = C.c_double(0.0), C.c_int64(0)
    if not lazy:
        sw_ms.value, sw_cnt.value = opt.sweep_launch_ms()
        opt.keep_sweep_events(False)
    executed, survey = step_flops(cfg)
    ms_step = dt / steps * 1e3
    if lazy:
        # no table sweep: what bounds the step is its matrix work; the HBM side is the touched rows only (7 x 4 x D
        # bytes per looked-up row -- reported separately, never mixed with the dense-exact figure)
        n_rows = cfg["B"] * (2 + (cfg["H"] if cfg["model"] != "base" else 0))
        roof = {"bound": "mfma", "kernel": "whole step (no sweep: logits + towers" + (" + encoder" if cfg["model"] != "base" else "") + ")",
                "achieved": round(executed / (ms_step * 1e-3) / 1e12, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(executed / (ms_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                "executed_flops_per_step": executed, "survey_flops_per_step": survey,
                "touched_rows_bytes_per_step": 28.0 * cfg["D"] * n_rows,
                "touched_rows_GBps": round(28.0 * cfg["D"] * n_rows / (ms_step * 1e-3) / 1e9, 1)}
    elif cfg["model"] != "base":
        # history model: SURVEY 8d's bound is the fp32 matrix pipe (the encoder), priced on executed AND on the survey's flops
        roof = {"bound": "mfma", "kernel": "whole step (encoder + towers + logits; the sweep runs underneath)",
                "achieved": round(executed / (ms_step * 1e-3) / 1e12, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(executed / (ms_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                "executed_flops_per_step": executed, "survey_flops_per_step": survey,
                "frac_on_survey_flops": round(survey / (ms_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)}
    else:
        # base model: the dense-exact table sweep (24 B per table element), per launch and per step
        nbytes = algorithmic_sweep_bytes(cfg, 1)
        launch_ms = sw_ms.value / sw_cnt.value if sw_cnt.value else None
        roof = {"bound": "hbm", "kernel": "adam_sweep_tables_kernel", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "achieved": round(nbytes / (launch_ms * 1e-3) / 1e9, 1) if launch_ms else None,
                "frac": round(nbytes / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if launch_ms else None,
                "avg_launch_ms": round(launch_ms, 4) if launch_ms else None, "launches": sw_cnt.value,
                "algorithmic_bytes_per_launch": nbytes, "traffic": None,
                "step_achieved": round(nbytes / (ms_step * 1e-3) / 1e9, 1),
                "step_frac": round(nbytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return {"workload": name + (" [value-exact DEFERRED Adam, K steps + flush: not the headline schedule]" if lazy else "")
                        + (" [whole step replayed as ONE hipGraph]" if graph else "")
                        + (" [fresh uniform ids every step]" if fresh_ids else (" [8 batches cycled: every row recurs after 8 steps]" if lazy else "")),
            "pairs_per_s": round(cfg["B"] * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
            "steps": steps, "warmup": warmup, "B": cfg["B"], "n_items": cfg["n_items"],
            "H": cfg["H"] if cfg["model"] != "base" else None, "roofline": roof,
            **({} if lazy else {"sweep_workgroups": opt._sweep_wgs or 768}),  # where the level scan of the optimizer settled
            **({"note": f"STEADY STATE of the deferred schedule: {warmup} untimed steps of fresh uniform ids first (N_i / B = "
                        f"{cfg['n_items'] // cfg['B']} = r; a looked-up item row then carries moments with probability "
                        f"{1 - math.exp(-(warmup + steps / 2) * cfg['B'] / cfg['n_items']):.2f} and is replayed over "
                        f"{_expected_idle(warmup + steps / 2, cfg['n_items'] / cfg['B']) / (cfg['n_items'] / cfg['B']):.2f} r idle steps on "
                        f"average -- the limit is 1.00 and 1.00 r), then {steps} timed steps; the final flush ({flush_s:.2f} s, once per "
                        "run) is reported here, not inside the per-step figure",
                "flush_seconds": round(flush_s, 3)} if (fresh_ids and steady and flush_s is not None) else
               {"note": f"rows have idled at most {steps} steps when they are replayed; the replay cost grows with the idle time"}
               if fresh_ids else {})}


def _timed_mips(device, lib, N, Cn=10_000_000, B=1024, K=1000, D=128, reps=20):
    """BASELINE config 5 on one GPU: BaselineMIPSModule top-K over a 10 M-row corpus, fp32 and bf16 storage;
    roofline of the dense scoring pass (2*B*C*D flops) from HIP events on its stream."""
    import ctypes as C
    import two_tower_models_amd as A
    g = torch.Generator(device=device).manual_seed(0)
    m = A.BaselineMIPSModule(corpus_size=8, embedding_dim=D)
    m.corpus = torch.randn(Cn, D, device=device, generator=g)
    m.corpus_size = Cn
    q = torch.randn(B, D, device=device, generator=g)
    out = {}
    try:
        out["cpu_baseline"] = cpu_baseline_mips(m.corpus.cpu(), K)
    except Exception as e:
        out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    for name, peak in (("fp32", MFMA_F32_PEAK_TF), ("fp32_split16", MFMA_BF16_PEAK_TF), ("bf16", MFMA_BF16_PEAK_TF)):
        if name == "fp32_split16":
            # EXPLORATORY, reported separately (never instead of "fp32"): the fp32 corpus scored from its two-term fp16 split,
            # three fp16 MFMA products per product (csrc/mips.hip TT_F16X2) -- same contract, fp32-grade scores
            m.use_split_fp16_scoring()
        if name == "bf16":
            m.use_split_fp16_scoring(False)
            m.use_bf16_storage()
        m.search(q, K)  # warm-up: allocates the workspace
        torch.cuda.synchronize()
        lib.tt_profile_filter(b"mips_score_kernel")
        lib.tt_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(reps):
            m.search(q, K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        ms, cnt = C.c_double(0), C.c_int64(0)
        N.check(lib.tt_profile_read(b"mips_score_kernel", C.byref(ms), C.byref(cnt)), "tt_profile_read")
        lib.tt_profile_enable(0)
        products = 3.0 if name == "fp32_split16" else 1.0  # MFMA products the pipe executes per product of the algorithm
        tf = products * 2.0 * B * Cn * D * cnt.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else None
        sustained = mfma_sustained_peak(lib, N, device, N.TT_F32 if name == "fp32" else N.TT_BF16)
        out[name] = {"queries_per_s": round(B / dt, 1), "ms_per_call": round(dt * 1e3, 3), "C": Cn, "B": B, "K": K, "D": D,
                     "roofline": {"bound": "mfma", "kernel": "mips_pass1_dma_kernel", "achieved": round(tf, 1) if tf else None,
                                  "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4) if tf else None,
                                  # the register-only MFMA loop of the same dtype, same process, right after these calls: what the
                                  # pipe sustains at this box's power budget (fp16 ~ bf16: same pipe, same rate)
                                  "sustained_peak": sustained, "frac_of_sustained": round(tf / sustained, 4) if (tf and sustained) else None,
                                  "avg_launch_ms": round(ms.value / max(cnt.value, 1), 4), "launches": cnt.value,
                                  "algorithmic_flops_per_launch": products * 2.0 * B * Cn * D}}
        if name == "fp32_split16":
            out[name]["EXPLORATORY"] = ("dtype f32 (fp16x2 split): opt-in BaselineMIPSModule.use_split_fp16_scoring(); roofline priced on "
                                        "the 3 fp16 MFMA products per product against the fp16 pipe's peak; fp32-equivalent "
                                        f"{round(tf / 3.0, 1) if tf else None} TFLOP/s; parity: tests/test_gpu_fullsize.py::"
                                        "test_mips_split_fp16_scoring_exact_arithmetic_and_random")
        # serving-size batch: back-to-back calls of 16 queries (the pass is the corpus stream there, not the matrix cores)
        q16 = q[:16].contiguous()
        m.search(q16, K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m.search(q16, K)
        torch.cuda.synchronize()
        dt16 = (time.perf_counter() - t0) / 10
        out[name]["B16_ms_per_call"] = round(dt16 * 1e3, 3)
        out[name]["B16_corpus_stream_GBps"] = round(Cn * D * (2 if name == "bf16" else 4) / dt16 / 1e9, 1)
    return out


def secondary(device, lib, N):
    """The other BASELINE configs in the driver-run record (each a few hundred ms of GPU time): C2 and C3
    train steps, config 5's MIPS at C = 10 M / K = 1000 (fp32, bf16), and the deferred-Adam figure, labelled."""
    sec = {}
    # first, straight after the headline and before anything fragments the allocator's pools (the 155 GB of C4 in
    # particular): this figure is compared with the headline to a few per cent
    try:
        sec["P_sharded_W1"] = _timed_sharded_w1(device)
    except Exception as e:
        sec["P_sharded_W1"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    # (dense-exact workloads: 80 warm-up steps, the sweep-level scan of DenseExactAdam._tune_sweep settles inside them)
    for key, name, steps, lazy, fresh in (("C2", "C2", 100, False, False), ("C3", "C3", 60, False, False),
                                          ("P_lazy", "P", 20, True, False),
                                          # the same schedule with the step replayed as one hipGraph: the eager loop is bound by
                                          # the HOST's ~90 launches per step, the GPU work is ~0.5 ms of MFMA
                                          ("P_lazy_graphed", "P", 40, True, "graph"),
                                          # the deferred schedule's STEADY STATE next to -- not instead of -- the recurring-ids
                                          # figure: new uniform ids every step, pre-aged (see below)
                                          ("P_lazy_fresh_ids", "P", 300, True, True)):
        try:
            # fresh ids: pre-aged over 4 N_i / B untimed steps, so the figure IS the steady state to within 10 % of the replay
            # work per lookup (the 200-step window flattered it 2.7x); see the note for the exact fractions
            graph = fresh == "graph"
            fresh = fresh is True
            sec[key] = _timed_train(name, device, steps, (5000 if fresh else 3) if lazy else 80, lazy=lazy, fresh_ids=fresh,
                                    steady=fresh, graph=graph)
            if not lazy:
                sec[key]["cpu_baseline"] = cpu_baseline_small(name)
        except Exception as e:  # a secondary figure must never take the headline line down with it
            sec[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    try:  # BASELINE config 1 (the reference's CPU-plumbing shape: 1 K users x 10 K items, d = 32, B = 128) on both sides
        sec["C1"] = _timed_train("tiny", device, 200, 100)
        sec["C1"]["cpu_baseline"] = cpu_baseline_small("tiny", n_steps=50)
    except Exception as e:
        sec["C1"] = {"error": f"{type(e).__name__}: {e}"}
    try:  # the same shape with the whole dense-exact step replayed as ONE hipGraph: C1 is bound by the host's ~40 launches per step
        sec["C1_graphed"] = _timed_train("tiny", device, 400, 20, graph=True)
        sec["C1_graphed"]["note"] = ("GraphedTrainStep (forward + backward + dense-exact Adam with its side-stream sweep as a graph "
                                     "branch), bit-identical to the eager step: tests/test_gpu_models.py::test_graphed_train_step_*")
    except Exception as e:
        sec["C1_graphed"] = {"error": f"{type(e).__name__}: {e}"}
    try:  # BASELINE config 4's table (100 M items, 155 GB of p, m, v) on ONE MI355X: what each of 8 ranks sweeps is 1/8 of it
        sec["C4_1gpu"] = _timed_train("C4", device, 6, 6)
    except Exception as e:
        sec["C4_1gpu"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    try:
        # what the 10 M pairs/s @ 8 GPUs target rests on: ONE rank's kernels of the W = 8 step,
        # stand-in collectives -- clearly labelled, with the two logits kernels' own rooflines and the host's cost
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import bench_emulated_world as _emu
        sec["emulated_W8"] = _emu.emulated(8, "P", steps=30, warmup=25, device=device)  # warm-up covers the trainer's 15-step schedule scan
    except Exception as e:
        sec["emulated_W8"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    for key, wl in (("emulated_W8_C3", "C3"), ("emulated_W8_C4", "C4")):
        try:
            # the other two sharded configs sized on the product path: config 3 (B*H = 204 800 history rows routed per rank and
            # step) and config 4 (12.5 M item rows per rank: sweep-bound again)
            sec[key] = _emu.emulated(8, wl, steps=20, warmup=25, device=device)
        except Exception as e:
            sec[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    try:
        # BASELINE config 5's 8-GPU form, one rank's kernels (C/8 corpus rows x all 8 x 1024 queries, candidate lists, merge)
        sec["emulated_W8_mips"] = _emu.emulated_mips(8, device=device)
    except Exception as e:
        sec["emulated_W8_mips"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    try:
        # EXPLORATORY x deferred schedule: P_lazy with the in-batch logits as split-fp16 products (TT_CE_F16X2 on the module path)
        from two_tower_models_amd import ops as _ops
        _was = _ops._CE_F16X2
        _ops._CE_F16X2 = True
        try:
            sec["P_lazy_f16x2"] = _timed_train("P", device, 20, 3, lazy=True)
            sec["P_lazy_f16x2"]["EXPLORATORY"] = ("dtype f32 (fp16x2 split logits: csrc/ce_f16x2.hip) on top of the value-exact deferred Adam "
                                                 "schedule; compare with P_lazy, never with the headline")
        finally:
            _ops._CE_F16X2 = _was
    except Exception as e:
        sec["P_lazy_f16x2"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    try:
        # EXPLORATORY, reported separately like P_lazy: the same emulated step with the logits pair as
        # split-fp16 products (csrc/ce_f16x2.hip, TT_CE_F16X2) -- fp32-grade results at about half the logits time
        sec["emulated_W8_f16x2"] = _emu.emulated(8, "P", steps=30, warmup=25, device=device, split16=True)
    except Exception as e:
        sec["emulated_W8_f16x2"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    try:
        # the drop-in training script end to end (ref:train/train.py:138-183): dataset generated in HBM, device-side
        # shuffle + batch slicing, 1-D [B] labels, 40 steps per epoch at the headline shapes; 2nd epoch reported
        import argparse as _ap
        from two_tower_models_amd import train as T
        a = T.build_parser().parse_args([
            "--num_users", "1000000", "--user_id_hash_size", "1000000", "--num_items", "10000000",
            "--item_id_hash_size", "10000000", "--embedding_dim", "128", "--feature_dim", "8", "--batch_size", "8192",
            "--num_samples", str(40 * 8192), "--num_epochs", "2", "--user_history_seqlen", "4"])
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            st = T.main(a)
        sec["train_py_end_to_end"] = {"workload": "python -m two_tower_models_amd.train at the P shapes (N_u=1 M, N_i=10 M, D=128, "
                                                  "B=8192), on-device DummyRecDataset + shuffled batches, [B] labels",
                                      "pairs_per_s": round(st[-1]["pairs_per_s"], 1),
                                      "ms_per_step": round(st[-1]["seconds"] / 40 * 1e3, 4), "steps_per_epoch": 40,
                                      "epochs": 2}
    except Exception as e:
        sec["train_py_end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    try:
        sec["C5_mips"] = _timed_mips(device, lib, N)
    except Exception as e:
        sec["C5_mips"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return sec


def timed_c5(cfg, device, world, rank, steps, warmup, sharded, comm_timing=True):
    """BASELINE config 5: `model.forward(user_id, user_features, user_history)` of TwoTowerWithDebiasing -> top-K = 1000 of a
    10 M-row bf16 corpus, B = 1024 users per rank.  Row-sharded (`sharded`): tables and corpus in W row blocks behind the
    same modules (parallel.row_sharded); per call one all-gather of the queries, a local top-K over C / W rows for W*B
    queries, one all-to-all of the [W, B, K] candidate lists, tt_mips_merge.  Returns this rank's record: wall seconds for
    `steps` calls, the dense scoring pass' HIP-event time (tt_profile), per-exchange times."""
    import ctypes as C
    from two_tower_models_amd import _native as N
    from two_tower_models_amd import collectives, parallel
    lib = N.load()
    mp = cfg["mips"]
    if sharded:
        with parallel.row_sharded():
            model = build_model(cfg, device, seed=1000 + rank)
        parallel.shard_model_(model)
    else:
        model = build_model(cfg, device)
    model.eval()
    batches = [b[:3] for b in make_batches(cfg, 4, device, seed=1234 + 1000 * rank)]

    def step(i):
        with torch.no_grad():
            return model(*batches[i % len(batches)])

    group = parallel.dist.get_world_size() if sharded else 1  # (tools/bench_emulated_world.py swaps parallel.dist for stand-ins)

    def barrier():
        if group > 1:
            parallel.dist.barrier()
        torch.cuda.synchronize()

    for i in range(max(warmup, 1)):
        top = step(i)
    assert top.shape == (cfg["B"], mp["K"]) and top.dtype == torch.int64
    barrier()
    lib.tt_profile_filter(b"mips_score_kernel")
    lib.tt_profile_enable(1)
    if sharded and comm_timing:
        collectives.comm_timing(True)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    comm_ms = collectives.comm_timing_summary(steps) if (sharded and comm_timing) else None
    collectives.comm_timing(False)
    ms, cnt = C.c_double(0), C.c_int64(0)
    N.check(lib.tt_profile_read(b"mips_score_kernel", C.byref(ms), C.byref(cnt)), "tt_profile_read")
    lib.tt_profile_enable(0)
    rows_local = model.mips_module.corpus.shape[0]
    n_q = cfg["B"] * group  # queries this rank scores per call (all ranks' queries on its own block)
    # one dense scoring launch covers at most 1024 queries (csrc/mips.hip MIPS_QBATCH): W*B queries = several launches per call
    flops_call = 2.0 * n_q * rows_local * cfg["D"]
    flops = flops_call * steps / max(cnt.value, 1)  # per launch
    tf = flops * cnt.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else None
    peak = MFMA_BF16_PEAK_TF if mp["bf16"] else MFMA_F32_PEAK_TF
    sustained = mfma_sustained_peak(lib, N, device, N.TT_BF16 if mp["bf16"] else N.TT_F32)
    roof = {"bound": "mfma", "kernel": "mips_pass1_dma_kernel", "achieved": round(tf, 1) if tf else None, "peak": peak,
            "unit": "TFLOP/s", "frac": round(tf / peak, 4) if tf else None,
            "sustained_peak": sustained, "frac_of_sustained": round(tf / sustained, 4) if (tf and sustained) else None,
            "avg_launch_ms": round(ms.value / max(cnt.value, 1), 4), "launches": cnt.value,
            "algorithmic_flops_per_launch": flops, "launches_per_call": round(cnt.value / max(steps, 1), 2),
            "scoring_ms_per_call": round(ms.value / max(steps, 1), 4), "traffic": None,
            "sustained_peak_note": "sustained_peak = tt_mfma_probe in this process (register-only MFMA loop, random operands: the "
                                   "pipe at this box's power budget, 0.58-0.66 of the 2.5 PF spec figure on this class of box)"}
    return {"seconds": dt, "ms_per_call": dt / steps * 1e3, "roofline": roof, "comm_ms": comm_ms,
            "comm_bytes": dict(parallel.comm_bytes), "corpus_rows_this_rank": rows_local, "queries_scored_per_call_this_rank": n_q,
            "corpus_dtype": str(model.mips_module.corpus.dtype).replace("torch.", ""), "model": model}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec this script as N ranks under
    torch.distributed.run (one rank per GPU, RCCL) on a free local port, pass the ranks' output through
    and print rank 0's JSON line LAST.  On a box with fewer than N devices (the 1-GPU test boxes) the
    ranks share cuda:0 and exchange through gloo -- bench.py's TT_DIST_BACKEND test hook -- and
    the JSON line says so in config.parallelism."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    if "TT_DIST_BACKEND" not in env and torch.cuda.device_count() < n:
        sys.stderr.write(f"bench.py: {torch.cuda.device_count()} device(s) < --gpus {n}: ranks share cuda:0 over gloo\n")
        env["TT_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, stdin=subprocess.DEVNULL)
    lines = r.stdout.splitlines()
    js = [i for i, l in enumerate(lines) if l.startswith('{"metric"')]
    for i, l in enumerate(lines):
        if not js or i != js[-1]:
            print(l)
    if js:
        print(lines[js[-1]], flush=True)
    return r.returncode if r.returncode != 0 else (0 if js else 1)


def main_c5(args, cfg, device, world, rank, dist_backend, use_sharded):
    """`bench.py --workload C5 [--gpus N]`: queries/s of TwoTowerWithDebiasing.forward() -> top-1000 of a 10 M-row bf16 corpus."""
    import torch.distributed as dist
    from two_tower_models_amd import collectives
    ident = None
    if use_sharded:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1")
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(dist_backend)
        if args.transport == "native" and world > 1 and dist_backend == "nccl":
            from two_tower_models_amd.comm import NativeComm
            collectives.use_native_transport(NativeComm.from_torch_distributed(device))
        if dist_backend == "nccl":
            ident = rccl_identity(device, world)
    rec = timed_c5(cfg, device, world, rank, args.steps, args.warmup, use_sharded)
    rec.pop("model")
    dt = rec["seconds"]
    rank_ms = None
    if world > 1:
        every = collectives.all_gather_rows(torch.tensor([dt], device=device, dtype=torch.float64))
        rank_ms = {"min": round(float(every.min()) / args.steps * 1e3, 4), "max": round(float(every.max()) / args.steps * 1e3, 4)}
        dt = float(every.max())
    if rank == 0:
        mp = cfg["mips"]
        out = {"metric": f"queries/sec (TwoTowerWithDebiasing.forward -> baseline_mips_module top-K={mp['K']} over a "
                         f"{mp['C'] // 1_000_000} M-row {'bf16' if mp['bf16'] else 'fp32'} corpus; BASELINE config 5)",
               "value": cfg["B"] * world * args.steps / dt, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if mp["bf16"] else "f32", "data": "synthetic",
               "config": {"workload": f"C5: user tower (N_u={cfg['n_users']}, N_i={cfg['n_items']}, D={cfg['D']}, H={cfg['H']} history "
                                      f"encoder) + MIPS top-{mp['K']} over C={mp['C']} rows, B={cfg['B']} users/GPU",
                          "global_batch": cfg["B"] * world,
                          "parallelism": "single GPU" if not use_sharded else
                          f"tables and corpus row-sharded x{world} behind the module API (C/{world} = {rec['corpus_rows_this_rank']} corpus "
                          f"rows per GPU, every GPU scores all {rec['queries_scored_per_call_this_rank']} queries on its block), "
                          + ("RCCL" if dist_backend == "nccl" else dist_backend)},
               "roofline": rec["roofline"], "n_ranks": world if use_sharded else 1,
               "dist_backend": dist.get_backend() if use_sharded else None, "rccl": ident,
               "comm": {"bytes_sent_per_rank_per_call": rec["comm_bytes"], "ms_per_call": rec["comm_ms"],
                        "exposed_ms_per_call_total": round(sum(v["exposed_ms"] for v in (rec["comm_ms"] or {}).values()), 4)}
               if use_sharded else None,
               "rank_ms_per_step": rank_ms, "cpu_baseline": None}
        result = json.dumps(out)
    if use_sharded:
        collectives.use_native_transport(None)
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(result, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup", type=int, default=40,
                    help="untimed steps per spin-up block BEFORE the --warmup ones (blocks repeat until two agree to 1 %%, at most 12; "
                         "0 = none): the first process on an idle box starts at cold clocks")
    ap.add_argument("--workload", default="P", choices=sorted(WORKLOADS),
                    help="P: the headline train step.  C5: BASELINE config 5's inference (queries/s of model.forward() -> top-1000 of "
                         "a 10 M-row bf16 corpus; --gpus N row-shards tables and corpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` object (C2 / C3 / deferred-Adam train steps, config 5 MIPS) the default "
                         "single-GPU P run appends")
    ap.add_argument("--overlap", default="forward", choices=["forward", "zero_grad", "off"],
                    help="when DenseExactAdam starts the table sweep (optim.py); all three are bit-identical")
    ap.add_argument("--sharded", action="store_true", help="use the row-sharded module path even at --gpus 1")
    ap.add_argument("--transport", default="torch", choices=["torch", "native"],
                    help="sharded collectives through torch.distributed's process group (default) or the C ABI's "
                         "tt_comm_* (RCCL bound by libtt_hotpath.so)")
    ap.add_argument("--check", action="store_true",
                    help="sharded runs: before the timed run, 3 steps at a small shape on all ranks; rank 0 repeats them on "
                         "the concatenated batch through the single-process module path and the losses must agree (1e-4)")
    ap.add_argument("--watchdog", type=float, default=30.0,
                    help="sharded runs: seconds without a completed step before the run is ended with the list of "
                         "exchanges in flight (0 = off)")
    ap.add_argument("--fresh-ids", action="store_true",
                    help="draw new uniform user / item ids on the device every step instead of cycling 16 batches "
                         "(the deferred schedule's steady state: every lookup hits rows that idled for ~N/B steps)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as one hipGraph (GraphedTrainStep; single-stream optimiser schedules)")
    ap.add_argument("--phase", default="step", choices=["step", "fwd", "fwdbwd"],
                    help="step: the metric (fwd + zero_grad + bwd + Adam).  fwd / fwdbwd: SURVEY 8d's secondary "
                         "figures -- forward only, or forward + backward without the optimiser step")
    ap.add_argument("--adam", default="dense", choices=["dense", "lazy"],
                    help="dense: the table sweep every step (the headline figure).  lazy: the value-exact deferred "
                         "schedule (SURVEY 8f-3, reported separately): K steps + a final flush inside the timed "
                         "region leave bit-identical tables")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # plain `python bench.py --gpus N`: become the launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # TT_DIST_BACKEND=gloo: test hook -- all ranks share cuda:0 and exchange through gloo (RCCL
    # refuses two ranks on one device), so the N > 1 code path can be exercised on a 1-GPU box.
    dist_backend = os.environ.get("TT_DIST_BACKEND", "nccl")
    device = torch.device(f"cuda:{local_rank if dist_backend == 'nccl' else 0}")
    torch.cuda.set_device(device)
    cfg = dict(WORKLOADS[args.workload])
    args.overlap = {"forward": "forward", "zero_grad": True, "off": False}[args.overlap]

    import two_tower_models_amd as A
    from two_tower_models_amd import _native as N
    lib = N.load()

    use_sharded = world > 1 or args.sharded
    n_ranks, dist_backend_seen = 1, None
    if cfg.get("mips"):  # BASELINE config 5: an inference workload with its own metric
        return main_c5(args, cfg, device, world, rank, dist_backend, use_sharded)
    if use_sharded and (args.adam != "dense" or args.fresh_ids or args.phase != "step" or args.graph):
        raise SystemExit("--adam lazy / --fresh-ids / --phase / --graph apply to the single-GPU module path")
    check = watchdog = None
    if use_sharded:
        import torch.distributed as dist
        from two_tower_models_amd import collectives, parallel
        if "MASTER_ADDR" not in os.environ:  # plain `python bench.py --sharded`
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(dist_backend)
        n_ranks, dist_backend_seen = dist.get_world_size(), dist.get_backend()
        if args.transport == "native" and (world > 1 or os.environ.get("TT_COMM_FORCE_ASYNC")):
            if dist_backend != "nccl":
                raise SystemExit("--transport native needs one GPU per rank (RCCL); this group runs over " + dist_backend)
            from two_tower_models_amd.comm import NativeComm
            collectives.use_native_transport(NativeComm.from_torch_distributed(device))
        if args.watchdog > 0:
            watchdog = collectives.Watchdog(args.watchdog)
        if args.check:
            check = sharded_check(device, rank, world)
            if rank == 0 and not check["ok"]:
                print(json.dumps({"metric": "bench.py --check FAILED", "check": check}), flush=True)
            if not check["ok"]:
                raise SystemExit(3)
        model, opt = build_sharded(cfg, device, rank)  # cfg['model'] selects base / hist / debias
        batches = make_batches(cfg, 16, device, seed=1234 + 1000 * rank)
        total_loss = torch.zeros((), device=device)
        step = sharded_step_fn(model, opt, total_loss, watchdog)
    else:
        model = build_model(cfg, device)
        opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep=args.overlap, lazy=args.adam == "lazy")
        batches = make_batches(cfg, 16, device)
        total_loss = torch.zeros((), device=device)

        if args.phase != "step":  # no optimiser step: nothing may start a sweep
            opt = A.DenseExactAdam(model.parameters(), lr=1e-3, overlap_sweep=False)

        id_gen = torch.Generator(device=device).manual_seed(4321)

        def refresh(batch):  # --fresh-ids: same features / labels, new uniform ids
            b = list(batch)
            b[0] = torch.randint(0, cfg["n_users"], tuple(b[0].shape), device=device, generator=id_gen)
            b[3] = torch.randint(0, cfg["n_items"], tuple(b[3].shape), device=device, generator=id_gen)
            if cfg["model"] in ("hist", "debias"):
                b[2] = torch.randint(0, cfg["n_items"], tuple(b[2].shape), device=device, generator=id_gen)
            return b

        pending = {}

        def step(batch, nxt=None):
            if args.fresh_ids:
                batch = pending.pop("next", None) or refresh(batch)
                if nxt is not None:
                    nxt = pending["next"] = refresh(nxt)
            if args.phase == "fwd":
                with torch.no_grad():
                    total_loss.add_(model.train_forward(*batch))
                return
            loss = model.train_forward(*batch)
            if args.adam == "lazy" and nxt is not None:  # replay the next batch's rows underneath this step
                opt.prefetch_rows(model._lookup_plan(nxt[0], nxt[2], nxt[3]))
            opt.zero_grad()
            loss.backward()
            if args.phase == "step":
                opt.step()
            else:
                for p in opt._tables:  # step() would have consumed these
                    p._tt_lookups.clear()
            total_loss.add_(loss.detach())  # the loop's loss accumulation, without the host sync

    if args.graph:
        if use_sharded or args.phase != "step":
            raise SystemExit("--graph applies to the single-GPU train step")
        graphed = A.GraphedTrainStep(model, opt, batches[0], warmup=2)

        def step(batch):  # noqa: F811 -- replaces the eager step
            total_loss.add_(graphed(*batch))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    single = not args.graph

    def run(i):
        if single:
            step(batches[i % len(batches)], batches[(i + 1) % len(batches)])
        else:
            step(batches[i % len(batches)])

    # Spin-up (untimed, before the contract's warm-up): blocks of --spinup steps until two consecutive blocks agree to 1 %,
    # at most 12 blocks.  The first process on a box that has sat idle runs its first second(s) of GPU work at a lower
    # memory clock: the same binary measured 5.77 / 5.93 ms in a window starting 0.3 s into the process and 5.39 / 5.44
    # ms for the same shape half a minute later in the same process (round 5).
    spun, last, spin_blocks_ms = 0, None, []
    while args.spinup > 0 and spun < 12 * args.spinup:
        torch.cuda.synchronize()
        t_blk = time.perf_counter()
        for i in range(args.spinup):
            run(spun + i)
        torch.cuda.synchronize()
        t_blk = time.perf_counter() - t_blk
        spun += args.spinup
        spin_blocks_ms.append(round(t_blk / args.spinup * 1e3, 4))  # -> the line's `spinup_ms_per_step_by_block`
        # (N > 1: every rank must run the SAME number of steps -- the collectives pair up -- so no rank-local decision: 2 blocks)
        if (world > 1 and spun >= 2 * args.spinup) or (world == 1 and last is not None and abs(t_blk - last) <= 0.01 * last):
            break
        last = t_blk
    for i in range(args.warmup):
        run(i)
    barrier()
    # HIP events around the kernels this line reports, and ONLY those (tt_profile_filter: an event pair changes what runs next
    # to what): the sweep / the deferred flush, and the backward logits kernel that takes over the roofline when it dominates
    # ... the sweep's own launch duration comes from the event pair the optimiser brackets it with anyway (keep_sweep_events)
    lib.tt_profile_filter(b"adam_flush_kernel" if args.adam == "lazy" else b"ce_bwd_kernel")
    lib.tt_profile_enable(1 if (args.adam == "lazy" or use_sharded) else 0)
    if args.adam != "lazy":
        opt.keep_sweep_events(True)
    if use_sharded:
        collectives.comm_timing(True)  # per-exchange events over the timed steps -> `comm.ms_per_step` below
    import gc
    gc.collect()
    gc.disable()  # no cyclic-GC pause inside the timed region (the loop allocates no cycles that would need it)
    clocks_before = sysfs_clocks()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run(args.warmup + i)
    if args.adam == "lazy" and not use_sharded:
        opt.flush()  # every deferred row update is paid for inside the timed region
    clocks_during = sysfs_clocks()  # the host has enqueued the steps and the GPU is still running them: clocks UNDER LOAD
    barrier()
    dt = time.perf_counter() - t0
    clocks_after = sysfs_clocks()
    gc.enable()
    comm_ms = None
    rank_ms = None
    if use_sharded:
        comm_ms = collectives.comm_timing_summary(args.steps)
        collectives.comm_timing(False)
        if watchdog is not None:
            watchdog.close()
    if world > 1:
        mine = torch.tensor([dt], device=device, dtype=torch.float64)
        every = collectives.all_gather_rows(mine)
        rank_ms = {"min": round(float(every.min()) / args.steps * 1e3, 4), "max": round(float(every.max()) / args.steps * 1e3, 4)}
        dt = float(every.max())

    import ctypes as C
    ms, cnt = C.c_double(0.0), C.c_int64(0)
    prof_kernel = b"adam_flush_kernel" if args.adam == "lazy" else b"adam_sweep_kernel"
    if args.adam == "lazy":
        N.check(lib.tt_profile_read(prof_kernel, C.byref(ms), C.byref(cnt)), "tt_profile_read")
    else:
        ms.value, cnt.value = opt.sweep_launch_ms()
        opt.keep_sweep_events(False)
    ce_ms, ce_cnt = C.c_double(0.0), C.c_int64(0)
    N.check(lib.tt_profile_read(b"ce_bwd_kernel", C.byref(ce_ms), C.byref(ce_cnt)), "tt_profile_read")
    lib.tt_profile_enable(0)

    ident = rccl_identity(device, world) if (use_sharded and dist_backend == "nccl") else None
    if rank == 0:
        B = cfg["B"]
        pairs = B * world * args.steps
        sweep_bytes_step = algorithmic_sweep_bytes(cfg, world)  # per GPU per step (one launch spans both tables)
        roof = None
        if args.adam == "lazy":
            # the flush reads and rewrites each table once (24 B/element) and replays `steps` updates per
            # element in registers: VALU-bound, so the HBM figure is informational only
            if cnt.value > 0 and ms.value > 0:
                roof = {"bound": "hbm", "kernel": "adam_flush_kernel", "achieved": round(sweep_bytes_step / (ms.value * 1e-3) / 1e9, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sweep_bytes_step / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": None, "launches": cnt.value, "avg_launch_ms": round(ms.value / cnt.value, 4),
                        "algorithmic_bytes_per_launch": sweep_bytes_step / cnt.value}
        elif cnt.value > 0 and ms.value > 0:
            achieved = sweep_bytes_step * args.steps / (ms.value * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            traffic = pmc_traffic_bytes(tpath, args.workload, world)
            # the name rocprofv3 reports for it (profiles/r01_kernel_stats_P_1gpu_final.csv)
            sweep_name = "adam_sweep_tables_kernel"
            roof = {"bound": "hbm", "kernel": sweep_name, "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "traffic_source": ("profiles/pmc_traffic.json (static: rocprofv3 --pmc passes of this command, "
                                       "committed; not re-measured in this run)" if traffic is not None else None),
                    "launches": cnt.value,
                    "avg_launch_ms": round(ms.value / cnt.value, 4),
                    "algorithmic_bytes_per_launch": sweep_bytes_step * args.steps / cnt.value}
            # the box's own streaming rate, measured here and now (a copy as large as the sweep's footprint, same process, same
            # allocator state, right after the timed steps), and the clocks on both sides of the timed region: a line whose
            # `frac` moved can be attributed -- `frac_of_copy` moves with the code, `hbm_copy_GBps` with the box
            try:
                cal = hbm_copy_calibration(lib, N, device, min(sweep_bytes_step, 40e9))
                roof.update(hbm_copy_GBps=cal["GBps"], hbm_copy_best_GBps=cal["best_GBps"],
                            frac_of_copy=round(achieved / cal["GBps"], 4), hbm_copy=cal)
            except Exception as e:  # noqa: BLE001
                roof["hbm_copy_GBps"] = f"unavailable ({type(e).__name__}: {e})"
            # (before / after are read with the device idle between launches -- the shader clock may already have dropped)
            roof["clocks"] = {"before": clocks_before, "during": clocks_during, "after": clocks_after}
            roof["table_arena"] = getattr(opt, "arena_note", None)  # candidate allocations timed at the first step, the one kept
            roof["sweep_workgroups"] = opt._sweep_wgs or 768
            roof["sweep_level_decided_by"] = opt.sweep_level_note()
            # With the tables sharded over many GPUs the sweep shrinks 1/N while the global-negative
            # logits grow N-fold: report whichever kernel family actually dominates the step.
            if ce_cnt.value > 0 and ce_ms.value > ms.value:
                n_neg = B * world if use_sharded else B
                # one gradient product per launch (dI); ce_bwd_kernel also recomputes its logits tile,
                # ce_bwd_kept_kernel reads the logits the forward kept (wide negative sets: N >= 4 M, ops.InBatchSoftmaxCE)
                kept = use_sharded and world >= 4
                flops = 2.0 * B * n_neg * cfg["D"]
                tf = flops * ce_cnt.value / (ce_ms.value * 1e-3) / 1e12
                roof = {"bound": "mfma", "kernel": "ce_bwd_kept_kernel" if kept else "ce_bwd_kernel",
                        "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF,
                        "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": None,
                        "launches": ce_cnt.value, "avg_launch_ms": round(ce_ms.value / ce_cnt.value, 4),
                        "algorithmic_flops_per_launch": flops}
        out = {
            "metric": ("user-item pairs/sec (in-batch softmax train step: fwd + zero_grad + bwd + dense-exact Adam)"
                       if args.phase == "step" else f"user-item pairs/sec, {args.phase} only (secondary figure, SURVEY 8d)")
                      + (" [value-exact DEFERRED Adam: K steps + flush; not the headline schedule]" if args.adam == "lazy" else ""),
            "value": pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_steps": spun,
            "spinup_ms_per_step_by_block": spin_blocks_ms,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: TwoTowerBaseRetrieval train step, N_u={cfg['n_users']}, "
                                   f"N_i={cfg['n_items']}, D={cfg['D']}, F={cfg['F']}, B={B}/GPU"
                                   + (f", H={cfg['H']} history encoder" if cfg['model'] in ('hist', 'debias') else "")
                                   + (", debias loss head" if cfg['model'] == 'debias' else ""),
                       "global_batch": B * world,
                       "parallelism": ("single GPU" + (", whole-step hipGraph" if args.graph else "")) if not use_sharded else
                       f"row-sharded tables x{world} behind the module API (parallel.py), global in-batch negatives, "
                       + ("RCCL" if dist_backend == "nccl" else dist_backend)},
            "roofline": roof,
            # what the collective library itself reports (1 when no process group was needed)
            "n_ranks": n_ranks, "dist_backend": dist_backend_seen, "rccl": ident,
        }
        # tuned kernels that did NOT run for this shape, each with the constraint that ruled it out ({} = all taken)
        from two_tower_models_amd import ops as _ops
        out["generic_paths"] = dict(_ops.generic_paths)
        if use_sharded:  # bytes each rank sends to its peers per step, by exchange (parallel.py)
            out["comm"] = {"transport": args.transport, "bytes_sent_per_rank_per_step": dict(parallel.comm_bytes),
                           "total_MB": round(sum(parallel.comm_bytes.values()) / 1e6, 2),
                           # rank 0, HIP events around every exchange of the timed steps, ms per step:  span = issue -> result
                           # usable (the cost if nothing overlapped it), exposed = how long the compute stream stood still
                           # at the wait (what the exchange actually adds to the step), wire = the collective alone (native
                           # transport only).  A bad 1 -> 8 curve is read off `exposed_ms` by exchange.
                           "ms_per_step": comm_ms,
                           "exposed_ms_per_step_total": round(sum(v["exposed_ms"] for v in (comm_ms or {}).values()), 4),
                           "schedule": {"table_sweep": "forward-announced (starts once the owners have served the step's lookups)",
                                        "sweep_workgroups": opt._sweep_wgs or 768, "sweep_level_decided_by": opt.sweep_level_note()}}
            out["rank_ms_per_step"] = rank_ms  # slowest / fastest rank's own clock over the timed steps (null at N = 1)
            out["check"] = check
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        else:
            out["cpu_baseline"] = None
        default_run = (world == 1 and not use_sharded and args.workload == "P" and args.phase == "step"
                       and args.adam == "dense" and not args.graph and not args.fresh_ids)
        if default_run and not args.no_secondary:
            # release the headline model's 17 GB first
            model = opt = batches = step = graphed = None  # noqa: F841
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["secondary"] = secondary(device, lib, N)
        result = json.dumps(out)
    if use_sharded:
        collectives.use_native_transport(None)
        torch.distributed.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio; flush that first so the JSON line
        # is the LAST line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(result, flush=True)


if __name__ == "__main__":
    main()
