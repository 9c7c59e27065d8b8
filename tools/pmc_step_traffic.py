"""Per-step HBM traffic of a whole train step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: python tools/pmc_step_traffic.py <fetch_dir> <write_dir> <out_json> [algorithmic_GB]
Steps are delimited by adam_advance_kernel (or the first adam_begin_ids_kernel after a sweep) dispatches; per kernel name the counters are summed over the complete steps
and divided by their number.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 for kernels that stream 16-B / lane
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950) -- reported next to the
uncorrected (FETCH_SIZE + WRITE_SIZE) * 1024 so both bounds are on the page."""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    # a step begins at adam_advance_kernel / adam_begin_ids_kernel; large lookups launch adam_begin_ids_kernel TWICE per step
    # (p plane, then m and v on the sweep's stream), so a begin kernel opens a step only if a sweep ran since the last one
    marks, swept = [], True
    for i, r in enumerate(rows):
        if "adam_sweep" in r[1]:
            swept = True
        elif ("adam_advance_kernel" in r[1] or "adam_begin_ids_kernel" in r[1]) and swept:
            marks.append(i)
            swept = False
    if len(marks) < 3:
        raise SystemExit(f"{d}: need >= 3 steps")
    a, b = marks[1], marks[-1]  # skip the first step (allocations, first-use set-up)
    steps = len(marks) - 2
    acc = collections.defaultdict(float)
    for _, name, v in rows[a:b]:
        acc[name.split("(")[0].replace("void ", "").replace("tt::", "")[:70]] += v / steps
    return acc, steps


fetch, ns = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
names = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))
kernels = {k: {"fetch_KB": round(fetch.get(k, 0), 1), "write_KB": round(write.get(k, 0), 1),
               "GB_corrected": round((2 * fetch.get(k, 0) + write.get(k, 0)) * 1024 / 1e9, 4)} for k in names}
tot_c = sum(v["GB_corrected"] for v in kernels.values())
tot_u = sum((fetch.get(k, 0) + write.get(k, 0)) * 1024 / 1e9 for k in names)
rec = {"steps_averaged": ns, "GB_per_step_corrected": round(tot_c, 3), "GB_per_step_uncorrected": round(tot_u, 3),
       "method": __doc__.split("usage:")[1].split("\n", 2)[2].strip(), "kernels": kernels}
if len(sys.argv) > 4:
    rec["algorithmic_GB_per_step"] = float(sys.argv[4])
    rec["corrected_over_algorithmic"] = round(tot_c / float(sys.argv[4]), 3)
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k != "kernels"}))
for k in names[:12]:
    print(f"  {k:70s} {kernels[k]['GB_corrected']:8.3f} GB")
