"""What decides the Adam table sweep's HBM rate on a given box?  (VERDICT r5 "next" item 2: the driver's box streamed the
P tables 8 % slower than the builder's while streaming C4's 155 GB at full rate in the same process.)

One process, every variant interleaved over several rounds on the SAME box, next to the box's own streaming-copy rate
and its clocks:
  * copy      tt_stream_copy of a P-sized buffer (16.9 GB read + 16.9 GB written), several widths
  * sweep     tt_adam_tables_sweep ALONE (no forward / backward next to it) over P-shaped tables (1 M + 10 M rows x 128),
              several widths, with the arrays placed
        torch     six separate allocations from torch's caching allocator (what DenseExactAdam does)
        arena     one allocation, the six arrays back to back (2 MiB-aligned)
        arena1g   one allocation, every array on its own 1 GiB boundary
        first     the arena allocated FIRST in the process (before anything else touched the device)
  * C4-sized sweep (100 M rows) when --big is given
Usage: python tools/sweep_placement.py [--big] [--rounds N]      (prints a table; bench.py is not involved)"""
import argparse
import ctypes as C
import glob
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sysfs_clocks():
    """Current sclk / mclk (MHz) of card 0 from the amdgpu sysfs files ('*' marks the active level); {} if unreadable."""
    out = {}
    for key, name in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk")):
        for path in sorted(glob.glob(f"/sys/class/drm/card*/device/{name}")):
            try:
                for line in open(path):
                    if "*" in line:
                        out[key] = int(line.split(":")[1].strip().split("M")[0])
                break
            except (OSError, ValueError, IndexError):
                continue
    for path in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")):
        try:
            out["power_W"] = round(int(open(path).read()) / 1e6, 1)
            break
        except (OSError, ValueError):
            continue
    return out


def box_info():
    info = {}
    for cmd in (["rocm-smi", "--showmemuse", "--showperflevel", "--showmaxpower", "--showmemorypartition", "--showcomputepartition"],
                ["rocm-smi", "--showclocks"]):
        try:
            info[" ".join(cmd[1:])] = subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout
        except Exception as e:  # noqa: BLE001
            info[" ".join(cmd[1:])] = f"{type(e).__name__}: {e}"
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from two_tower_models_amd import _native as N
    lib = N.load()
    rows = (1_000_000, 10_000_000)
    D = 128
    sizes = [r * D for r in rows]  # floats per array
    total_floats = 3 * sum(sizes)

    # ---- "first": before anything else allocates
    def carve(buf, align_bytes):
        """Six views (p, m, v of each table) of `buf`, each starting on a multiple of align_bytes."""
        base = buf.data_ptr()
        off = (-base) % align_bytes
        out = []
        for n in sizes:
            t = []
            for _ in range(3):
                t.append(buf[off // 4: off // 4 + n])
                off += n * 4
                off += (-(base + off)) % align_bytes
            out.append(t)
        return out

    first_buf = torch.empty(total_floats + (64 << 20), dtype=torch.float32, device=dev)
    placements = {"first": carve(first_buf, 2 << 20)}
    print("clocks at start:", sysfs_clocks(), flush=True)
    for k, v in box_info().items():
        print(f"--- rocm-smi {k}\n{v}", flush=True)
    placements["torch"] = [[torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3)] for n in sizes]
    arena = torch.empty(total_floats + (64 << 20), dtype=torch.float32, device=dev)
    placements["arena"] = carve(arena, 2 << 20)
    arena1g = torch.empty(total_floats + 7 * (1 << 28), dtype=torch.float32, device=dev)
    placements["arena1g"] = carve(arena1g, 1 << 30)
    for name, tabs in placements.items():
        for t in tabs:
            t[0].normal_()
            t[1].normal_().mul_(0.01)
            t[2].normal_().abs_().mul_(0.01)
        print(f"{name:8s} bases mod 1 GiB:", [[hex(x.data_ptr() % (1 << 30)) for x in t] for t in tabs], flush=True)
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 5, 0, 0, 0], dtype=torch.float64, device=dev)
    N.check(lib.tt_adam_advance(hyper.data_ptr(), N.stream()), "adv")

    def sweep(tabs, wgs, reps):
        descs = (N.AdamTensor * len(tabs))()
        for i, (p, m, v) in enumerate(tabs):
            descs[i].p, descs[i].m, descs[i].v, descs[i].n = p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            N.check(lib.tt_adam_tables_sweep(descs, len(tabs), hyper.data_ptr(), wgs, N.stream()), "sweep")
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in evs)

    def copy(wgs, reps):
        src, dst = first_buf[: total_floats // 2], arena[: total_floats // 2]
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            N.check(lib.tt_stream_copy(src.data_ptr(), dst.data_ptr(), src.numel() * 4, N.stream()), "copy")
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in evs), 2.0 * src.numel() * 4

    sweep_bytes = 24.0 * sum(sizes)
    widths = (256, 384, 512, 768, 1024)
    res = {}
    t0 = time.time()
    for rnd in range(args.rounds):
        for w in widths:
            ms, nbytes = copy(w, 6)
            res.setdefault(("copy", w), []).extend(ms[:-1])
            for name, tabs in placements.items():
                res.setdefault((name, w), []).extend(sweep(tabs, w, 6)[:-1])
        print(f"round {rnd}: {time.time() - t0:.0f} s, clocks {sysfs_clocks()}", flush=True)
    print(f"\nP-sized (sweep {sweep_bytes / 1e9:.2f} GB per launch; copy {nbytes / 1e9:.2f} GB per launch), median / min GB/s:")
    print(f"{'':10s}" + "".join(f"{w:>16d}" for w in widths))
    for name in ["copy"] + list(placements):
        nb = nbytes if name == "copy" else sweep_bytes
        row = ""
        for w in widths:
            v = sorted(res[(name, w)])
            row += f"{nb / v[len(v) // 2] / 1e6:>9.0f}/{nb / v[0] / 1e6:<6.0f}"
        print(f"{name:10s}{row}")
    if args.big:
        del placements, first_buf, arena, arena1g
        torch.cuda.empty_cache()
        n = 100_000_000 * D
        big = [[torch.empty(n, dtype=torch.float32, device=dev) for _ in range(3)]]
        for t in big[0]:
            t.zero_()
        for w in (512, 768):
            v = sweep(big, w, 4)
            print(f"C4-sized sweep (100 M rows, {24.0 * n / 1e9:.1f} GB), {w} workgroups: median {24.0 * n / v[len(v) // 2] / 1e6:.0f} GB/s")
    print("clocks at end:", sysfs_clocks())


if __name__ == "__main__":
    main()
