"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: for the LAST complete step
(delimited by adam_advance_kernel or adam_begin_ids_kernel, the first launch of a step) print every kernel's start / end relative to the step start.
usage: python tools/timeline.py <kernel_trace.csv> [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?")))
             for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if "adam_advance_kernel" in e[2] or "adam_begin_ids_kernel" in e[2]]
if marks:  # steps that park a big lookup launch the begin kernel twice (p plane on the main stream, m / v planes on the sweep's):
    q0 = ev[marks[0]][3]  # count the stream of the first one seen only
    marks = [i for i in marks if ev[i][3] == q0]
if len(marks) < 3:
    raise SystemExit("need >= 3 steps in the trace")
a, b = marks[-3], marks[-2]
t0 = ev[a][0]
print(f"step span {(ev[b][0] - t0) / 1e3:.1f} us, {b - a} kernels")
for s, e, name, q in ev[a:b]:
    if (e - s) / 1e3 >= min_us:
        short = name.split("(")[0].replace("void ", "").replace("tt::", "")[:60]
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q={q}  {short}")
