#!/bin/bash
# SQ counters of the split-fp16 logits kernels (where do their wave cycles go?).  Run on the GPU box; prints one line per
# counter and kernel, averaged over the dispatches of tools/bench_ce.py 8192 65536.   tools/ce16_pmc.sh [out_dir]
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/ce16_pmc}
mkdir -p $OUT
OUT=$(cd $OUT && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/p1 -- python $R/tools/bench_ce.py 8192 65536 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM \
  --kernel-trace --output-format csv -d $OUT/p2 -- python $R/tools/bench_ce.py 8192 65536 > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d $OUT/p3 -- python $R/tools/bench_ce.py 8192 65536 > $OUT/p3.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("ce16_fwd_kernel", "ce16_bwd_items_kernel", "ce_fwd_du_kernel", "ce_bwd_kept_kernel"):
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:24s} {k[1]:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
find $OUT -name "*.csv" -size +5M -delete; find $OUT -name "*.db" -delete
