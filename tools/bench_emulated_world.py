"""Per-rank COMPUTE of the row-sharded train step at world size W, on ONE MI355X: torch.distributed
is replaced by a local stand-in whose collectives produce tensors of the right shape without moving
data between processes (all_gather = W copies of the local block, reduce_scatter = the first block,
all_reduce = identity).  The numbers are NOT a training run -- they show what one rank's kernels cost
once the tables are W times thinner and the in-batch negatives W times wider, i.e. the step time at
W GPUs minus the collectives.  bench.py puts `emulated(8, "P")` into the default line's `secondary`.
Usage: python tools/bench_emulated_world.py [W] [workload]"""
import ctypes as C
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32 MFMA peak


class _Done:
    def wait(self):
        return True


def _fake_dist(W):
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

    return types.SimpleNamespace(
        all_to_all_single=_all_to_all, get_backend=lambda: "emulated", get_world_size=lambda: W, get_rank=lambda: 0,
        is_initialized=lambda: True, all_gather_into_tensor=_all_gather, reduce_scatter_tensor=_reduce_scatter,
        all_reduce=lambda x, op=None, async_op=False: (_Done() if async_op else None),
        broadcast=lambda x, src=0: None, ReduceOp=torch.distributed.ReduceOp, barrier=lambda: None)


def _time_steps(step, batches, steps):
    """-> (ms to ENQUEUE a step, ms per step): wall clock around the enqueue loop, then around the drain."""
    n = len(batches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(batches[i % n], batches[(i + 1) % n])
    host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    return host, (time.perf_counter() - t0) / steps * 1e3


MFMA_F16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA peak


def emulated(W=8, workload="P", steps=30, warmup=25, device=None, verbose=False, split16=None):
    """One rank's kernels of the W-GPU step (stand-in collectives) THROUGH THE MODULE PATH (parallel.row_sharded +
    TwoTower* + DenseExactAdam -- the code `bench.py --gpus W` runs), as a record for bench.py's `secondary`.
    split16: run the logits pair as split-fp16 products (csrc/ce_f16x2.hip, exploratory; default: what TT_CE_F16X2 says)."""
    import bench
    from two_tower_models_amd import _native as N
    from two_tower_models_amd import collectives, ops, parallel
    device = device or torch.device("cuda:0")
    lib = N.load()
    real = (collectives.dist, parallel.dist, ops._CE_F16X2)
    collectives.dist = parallel.dist = _fake_dist(W)
    if split16 is not None:
        ops._CE_F16X2 = bool(split16)
    split = ops._CE_F16X2
    try:
        cfg = dict(bench.WORKLOADS[workload])
        # (1) the host's own cost of enqueueing a step -- Python + ~100 launches -- measured where the GPU can never
        # be what the enqueue loop waits for: the SAME step on tables and batches 64 times smaller (every kernel of
        # the full-size step is launched, each finishes in microseconds, the queue never fills)
        tiny_cfg = dict(cfg, n_users=max(cfg["n_users"] // 64, 1024), n_items=max(cfg["n_items"] // 64, 1024), B=128)
        model, opt = bench.build_sharded(tiny_cfg, device, 0)
        tb = bench.make_batches(tiny_cfg, 8, device)
        tstep = bench.sharded_step_fn(model, opt, torch.zeros((), device=device))
        for i in range(warmup):
            tstep(tb[i % 8], tb[(i + 1) % 8])
        host_only_ms, _ = _time_steps(tstep, tb, steps)
        del model, opt, tb, tstep
        torch.cuda.empty_cache()

        model, opt = bench.build_sharded(cfg, device, 0)
        batches = bench.make_batches(cfg, 8, device)
        step = bench.sharded_step_fn(model, opt, torch.zeros((), device=device))
        if verbose and os.environ.get("EMU_TRACE"):  # measurement aid: wall clock per 10 steps while the controller scans
            real_sweep, seen = lib.tt_adam_tables_sweep, []

            def spy(descs, n, hyper, wgs, stream):
                seen.append(int(wgs))
                return real_sweep(descs, n, hyper, wgs, stream)

            lib.tt_adam_tables_sweep = spy
            for blk in range(max(warmup, 230) // 10):
                seen.clear()
                ms10 = _time_steps(step, batches, 10)[1]
                recs = [r for r in opt._tune_done[-10:] if r[2] is not None]
                sw = sorted(r[1].elapsed_time(r[2]) for r in recs)
                lag = sorted(r[0].elapsed_time(r[1]) for r in recs)  # step begin -> sweep start
                print(f"  steps {10 * blk + 1}-{10 * blk + 10}: {ms10:.3f} ms/step, level {opt._sweep_wgs or 768}, phase {opt._tune_state['phase']}, "
                      f"sweep launches {sorted(set(seen))}, median sweep {sw[len(sw) // 2] if sw else -1:.3f} ms starting {lag[len(lag) // 2] if lag else -1:.3f} ms into the step")
            lib.tt_adam_tables_sweep = real_sweep
        else:
            for i in range(max(warmup, 230)):  # covers the optimiser's group-wide sweep-level scan (optim._tune_sweep_group)
                step(batches[i % 8], batches[(i + 1) % 8])
        lib.tt_profile_filter(b"ce_fwd_kernel,ce_bwd_kernel,adam_sweep_kernel")
        lib.tt_profile_enable(0 if os.environ.get("EMU_NO_PROF") else 1)
        host_ms, ms = _time_steps(step, batches, steps)
        if verbose and os.environ.get("EMU_FORCE_LEVELS"):  # measurement aid: the same process, the level pinned by hand
            for lv in (int(v) for v in os.environ["EMU_FORCE_LEVELS"].split(",")):
                opt._tune_state = {"phase": os.environ.get("EMU_FORCE_PHASE", "locked"), "since": 0, "why": f"pinned to {lv}"}
                opt._tune_done.clear()
                opt._sweep_wgs = lv
                for i in range(int(os.environ.get("EMU_FORCE_WARM", "40"))):
                    step(batches[i % 8], batches[(i + 1) % 8])
                print(f"  pinned to {lv or 768} workgroups: " + ", ".join(f"{_time_steps(step, batches, 10)[1]:.3f}" for _ in range(5)) + " ms/step (5 x 10 steps)")
                recs = [r for r in opt._tune_done[-10:] if r[2] is not None]
                if recs:
                    sw = sorted(r[1].elapsed_time(r[2]) for r in recs)
                    lag = sorted(r[0].elapsed_time(r[1]) for r in recs)
                    print(f"    median sweep {sw[len(sw) // 2]:.3f} ms starting {lag[len(lag) // 2]:.3f} ms into the step")
        prof = {}
        for name in (b"ce_fwd_kernel", b"ce_bwd_kernel", b"adam_sweep_kernel"):
            t, c = C.c_double(0.0), C.c_int64(0)
            N.check(lib.tt_profile_read(name, C.byref(t), C.byref(c)), "tt_profile_read")
            prof[name.decode()] = (t.value / max(c.value, 1), c.value)
        lib.tt_profile_enable(0)

        # where the time goes: events on the main stream around the two logits Functions
        marks = []

        def _ev():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)

        _fwd, _bwd = ops.InBatchSoftmaxCE.forward, ops.InBatchSoftmaxCE.backward

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

        ops.InBatchSoftmaxCE.forward, ops.InBatchSoftmaxCE.backward = staticmethod(ce_fwd), staticmethod(ce_bwd)
        acc = [0.0] * 5
        try:
            # ten steps enqueued back to back, ONE synchronisation at the end: with a synchronisation per step the GPU starts
            # every step with an empty queue and the front of the step (routing, towers) runs at the HOST's pace -- that is what
            # rounds 3-5 reported as "lookups_towers 0.42-0.49 ms" (tools/emu_front_probe.py: GPU time per call = host time per call)
            per_step = []
            for i in range(12):
                marks.clear()
                _ev()
                step(batches[i % 8], batches[(i + 1) % 8])
                _ev()
                per_step.append(list(marks))
            torch.cuda.synchronize()
            for m in per_step[2:]:
                for k in range(5):
                    acc[k] += m[k].elapsed_time(m[k + 1]) / 10
        finally:
            ops.InBatchSoftmaxCE.forward, ops.InBatchSoftmaxCE.backward = staticmethod(_fwd), staticmethod(_bwd)
        B, D = cfg["B"], cfg["D"]
        M, Nn = B, B * W
        kept = Nn >= 4 * M and not split  # ops.InBatchSoftmaxCE: wide negative sets keep their logits for the backward
        # logit-sized products: forward S = U.I^T and E = P.I (2), backward dI = G^T.U from the kept logits (1) or
        # with S recomputed (2); 2*M*N*D flops each
        fl_fwd, fl_bwd = 4.0 * M * Nn * D, (2.0 if kept else 4.0) * M * Nn * D
        # the same two kernels with the GPU to themselves (no sweep next to them, nothing else in the queue): what the
        # kernels reach vs what the step gets out of them
        alone = {}
        split = split and bool(lib.tt_ce16_supported(M, Nn, D))
        if kept and not split:
            U = torch.randn(M, D, device=device) * 0.3
            I = torch.randn(Nn, D, device=device) * 0.3
            coef = torch.rand(M, device=device) / M
            lse, ce = torch.empty(M, device=device), torch.empty(M, device=device)
            du, dI = torch.empty(M, D, device=device), torch.empty(Nn, D, device=device)
            wsn = lib.tt_inbatch_ce_workspace_bytes(M, Nn, D)
            ws = torch.empty(wsn, dtype=torch.uint8, device=device)
            zn = lib.tt_inbatch_ce_logits_bytes(M, Nn)
            Z = torch.empty(zn, dtype=torch.uint8, device=device)

            def k_fwd():
                N.check(lib.tt_inbatch_ce_fwd_du_keep(U.data_ptr(), D, I.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), ce.data_ptr(),
                                                      du.data_ptr(), D, Z.data_ptr(), zn, ws.data_ptr(), wsn, N.stream()), "fwd_du_keep")

            def k_bwd():
                N.check(lib.tt_inbatch_ce_bwd_kept(U.data_ptr(), D, M, Nn, D, 0, lse.data_ptr(), coef.data_ptr(), Z.data_ptr(), zn,
                                                   dI.data_ptr(), D, ws.data_ptr(), wsn, N.stream()), "bwd_kept")

            for fn, key in ((k_fwd, "ce_fwd_kernel"), (k_bwd, "ce_bwd_kernel")):
                fn()
                torch.cuda.synchronize()
                lib.tt_profile_enable(1)
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                t, c = C.c_double(0.0), C.c_int64(0)
                N.check(lib.tt_profile_read(key.encode(), C.byref(t), C.byref(c)), "tt_profile_read")
                lib.tt_profile_enable(0)
                alone[key] = t.value / max(c.value, 1)
            del U, I, Z, ws, dI, du
        roof = []
        for kname, fl, key in (("ce16_fwd_kernel" if split else "ce_fwd_du_kernel", fl_fwd, "ce_fwd_kernel"),
                               (("ce16_bwd_items_kernel" if ops._CE16_KEEP else "ce16_bwd_items_rc_kernel") if split
                                else "ce_bwd_kept_kernel" if kept else "ce_bwd_kernel",
                                (2.0 if ops._CE16_KEEP else 4.0) * M * Nn * D if split else fl_bwd, "ce_bwd_kernel")):
            avg_ms, launches = prof[key]
            if launches and split:
                # every logit-sized product runs as THREE fp16 MFMA products: priced against the fp16 matrix-pipe peak on the
                # flops the pipe actually executes, next to the fp32-equivalent rate
                tf = 3.0 * fl / (avg_ms * 1e-3) / 1e12
                roof.append({"bound": "mfma", "kernel": kname, "achieved": round(tf, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                             "frac": round(tf / MFMA_F16_PEAK_TF, 4), "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                             "algorithmic_flops_per_launch": 3.0 * fl, "fp32_equivalent_TFLOPs": round(tf / 3.0, 1)})
            elif launches:
                tf = fl / (avg_ms * 1e-3) / 1e12
                roof.append({"bound": "mfma", "kernel": kname, "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                             "frac": round(tf / MFMA_F32_PEAK_TF, 4), "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                             "algorithmic_flops_per_launch": fl,
                             **({"alone_avg_launch_ms": round(alone[key], 4), "alone_frac": round(fl / (alone[key] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                                 "alone_note": "the same kernel at the same shape with the GPU to itself (10 launches back to back): the "
                                               "difference to `frac` is what sharing HBM and CUs with the Adam sweep costs it in the step"}
                                if key in alone else {})})
        # the table sweep of this rank's row blocks (HBM-bound; what the step is at C4's 12.5 M rows per rank)
        sw_bytes = bench.algorithmic_sweep_bytes(cfg, W)
        sw_ms = prof["adam_sweep_kernel"][0]
        sweep_roof = {"bound": "hbm", "kernel": "adam_sweep_tables_kernel", "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                      "achieved": round(sw_bytes / (sw_ms * 1e-3) / 1e9, 1) if sw_ms > 0 else None,
                      "frac": round(sw_bytes / (sw_ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4) if sw_ms > 0 else None,
                      "avg_launch_ms": round(sw_ms, 4), "launches": prof["adam_sweep_kernel"][1],
                      "algorithmic_bytes_per_launch": sw_bytes, "frac_of_step": round(sw_ms / ms, 3)}
        # padded fixed-capacity all-to-all vs the exact row bytes (SURVEY section 7 "History all-to-all volume")
        n_lookups = B * (2 + (cfg["H"] if cfg["model"] != "base" else 0))
        exact = (W - 1) / W * n_lookups * D * 4
        padded = parallel.comm_bytes.get("lookup_rows_alltoall", 0)
        pad = {"exact_bytes": int(exact), "padded_bytes": int(padded), "ratio": round(padded / exact, 3) if exact else None,
               "note": "cap = the largest (requester, owner) bucket over all ranks, rounded up to 64; ids are uniform here"}
        out = {
            "what": f"ONE rank's kernels of the row-sharded step at W = {W} on one GPU: tables 1/{W} as thick, {W}x{B} in-batch "
                    "negatives per user, routed lookups; torch.distributed replaced by stand-ins that return tensors of the right "
                    "shape without moving data -- NOT a training run, the collectives' time comes on top",
            "workload": workload, "world": W, "path": "module classes + DenseExactAdam over parallel.row_sharded tables",
            "kept_logits": (ops._CE16_KEEP if split else kept),
            "dtype": "f32 (fp16x2 split: every logits product as three fp16 MFMA products of two-term splits, fp32 accumulate)" if split else "f32",
            **({"EXPLORATORY": "TT_CE_F16X2 -- not the default path, not the headline; parity: tests/test_gpu_kernels.py::"
                               "test_split_fp16_ce_pair_vs_float64, tests/test_gpu_fullsize.py::test_split_fp16_ce_pair_8_rank_shape"}
               if split else {}),
            "ms_per_step_per_rank": round(ms, 4),
            "pairs_per_s_if_collectives_were_free": round(B * W / ms * 1e3, 1),
            "host_enqueue_ms_per_step": round(host_ms, 4),
            "host_only_enqueue_ms_per_step": round(host_only_ms, 4),
            "host_note": "host_only: the same step on 64x smaller tables / B = 128 (the GPU is never the wait); the difference to "
                         "host_enqueue is queue back-pressure, not Python",
            "main_stream_phases_ms": {"lookups_towers": round(acc[0], 3), "logits_fwd_dU": round(acc[1], 3), "weights_loss": round(acc[2], 3),
                                      "logits_bwd_dI": round(acc[3], 3), "towers_bwd_row_adam": round(acc[4], 3)},
            "roofline": roof,
            "sweep_avg_launch_ms": round(prof["adam_sweep_kernel"][0], 4),
            "sweep_roofline": sweep_roof,
            "lookup_rows_padding": pad,
            "sweep_level": opt.sweep_level_note(),
            "bytes_this_rank_would_send_per_step": dict(parallel.comm_bytes),
            "total_MB_sent": round(sum(parallel.comm_bytes.values()) / 1e6, 2),
            "steps": steps, "warmup": warmup,
        }
        if verbose:
            print(f"  host enqueue time {host_ms:.3f} ms/step (host only, GPU never the wait: {host_only_ms:.3f})")
            print("  main-stream phases (ms): lookups+towers %.3f | logits fwd + dU %.3f | weights/loss %.3f | logits bwd (dI) %.3f | "
                  "towers bwd + row Adam %.3f" % tuple(acc))
            print(f"  sweep level: {opt.sweep_level_note()}")
            print(f"emulated W={W} workload={workload}: {ms:.3f} ms/step per rank (no collectives) -> "
                  f"{B * W / ms * 1e3 / 1e6:.2f} M pairs/s if the collectives were free")
            for r in roof:
                print(f"  {r['kernel']}: {r['avg_launch_ms']:.3f} ms = {r['achieved']} TFLOP/s = {r['frac']:.3f} of the {'fp16' if r['peak'] > 1000 else 'fp32'} MFMA peak"
                      + (f" (alone: {r['alone_avg_launch_ms']:.3f} ms = {r['alone_frac']:.3f})" if "alone_frac" in r else ""))
            print(f"  bytes this rank would send per step: {parallel.comm_bytes} = {sum(parallel.comm_bytes.values()) / 1e6:.1f} MB")
        return out
    finally:
        collectives.dist, parallel.dist, ops._CE_F16X2 = real


def emulated_mips(W=8, steps=6, warmup=2, device=None, verbose=False):
    """ONE rank's kernels of BASELINE config 5 at W GPUs (bench.py --workload C5 --gpus W), stand-in collectives: this rank's
    C / W corpus rows (bf16) scored against ALL W x 1024 queries, the [W, B, K] candidate lists through the stand-in
    all-to-all, tt_mips_merge over W x K candidates per own query -- through TwoTowerWithDebiasing.forward() on a
    row-sharded model (tables and corpus)."""
    import bench
    from two_tower_models_amd import collectives, parallel
    device = device or torch.device("cuda:0")
    real = (collectives.dist, parallel.dist)
    collectives.dist = parallel.dist = _fake_dist(W)
    try:
        cfg = dict(bench.WORKLOADS["C5"])
        rec = bench.timed_c5(cfg, device, W, 0, steps, warmup, sharded=True, comm_timing=False)
        model = rec.pop("model")
        mp, B = cfg["mips"], cfg["B"]
        # the reference's full 3-tuple once (embeddings fetched through the routed exchange), timed separately: forward()
        # of the model discards them (ref:src/two_tower_base_retrieval.py:246-248)
        with torch.no_grad():
            q = torch.randn(B, cfg["D"], device=device)
            model.mips_module(query_embedding=q, num_items=mp["K"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.mips_module(query_embedding=q, num_items=mp["K"])
            torch.cuda.synchronize()
            full_ms = (time.perf_counter() - t0) * 1e3
            fetch_bytes = dict(parallel.comm_bytes)
        out = {"what": f"ONE rank's kernels of config 5 at W = {W} on one GPU: C/{W} = {rec['corpus_rows_this_rank']} {rec['corpus_dtype']} corpus "
                       f"rows scored against all {rec['queries_scored_per_call_this_rank']} queries, [W, B, K] candidate lists through a stand-in "
                       "all-to-all, tt_mips_merge over W x K candidates per own query, via TwoTowerWithDebiasing.forward() on a "
                       "row-sharded model -- NOT a serving run, the collectives' time comes on top",
               "world": W, "B_per_rank": B, "K": mp["K"], "C": mp["C"], "ms_per_call_per_rank": round(rec["ms_per_call"], 4),
               "queries_per_s_if_collectives_were_free": round(B * W / rec["ms_per_call"] * 1e3, 1),
               "roofline": rec["roofline"], "bytes_this_rank_would_send_per_call": rec["comm_bytes"],
               "total_MB_sent": round(sum(rec["comm_bytes"].values()) / 1e6, 2),
               "module_forward_3tuple_ms": round(full_ms, 3),
               "module_forward_3tuple_note": "BaselineMIPSModule.forward incl. the [B, K, D] embeddings fetched from their owners "
                                             "(routed all-to-all, fp32 rows on the wire): " + str(fetch_bytes),
               "steps": steps, "warmup": warmup}
        if verbose:
            print(json.dumps(out, indent=1))
        del model
        return out
    finally:
        collectives.dist, parallel.dist = real


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "C5":
        emulated_mips(int(sys.argv[1]), verbose=True)
        sys.exit(0)
    emulated(int(sys.argv[1]) if len(sys.argv) > 1 else 8, sys.argv[2] if len(sys.argv) > 2 else "P", verbose=True)
