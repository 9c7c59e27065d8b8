#!/bin/bash
# SQ counters of one kernel (where do its wave cycles go?).  Run on the GPU box; prints one line per counter, averaged
# over the matching dispatches.   tools/kernel_pmc.sh <kernel-name-substring> <out_dir> -- <command ...>
R=$(cd "$(dirname "$0")/.." && pwd)
PAT=$1; OUT=$2; shift 3
mkdir -p $OUT
OUT=$(cd $OUT && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/p1 -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d $OUT/p2 -- "$@" > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM \
  --kernel-trace --output-format csv -d $OUT/p3 -- "$@" > $OUT/p3.log 2>&1
python - $OUT "$PAT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
