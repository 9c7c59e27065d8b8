#!/bin/bash
# Measurement variants of the split-fp16 logits pair (ce_f16x2.hip -DTT_CE16_EXP=k) next to the product library, timed with
# tools/bench_ce.py at the W = 8 shape.  Variant results are WRONG by design.
#   1 no logits stores   2 no E product (forward)   4 non-temporal logits stores   8 tile wait leaves the four logits stores in flight   32 backward without logits loads
#   16 logits stores as four fully coalesced 1-KiB pieces per tile (wrong layout: what the 32-byte-per-row pattern costs)
#   256 logits stores straight from the score registers (32 bytes into each of 32 rows per instruction) instead of through the LDS transpose
#   128 forward operand reads where hipcc puts them (one MFMA ahead of their use) instead of a whole k-step / fragment pair ahead
#   tools/ce16_variants.sh build   (here)        tools/ce16_variants.sh run   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
P=two_tower_models_amd
mkdir -p $P/lib/exp
if [ "$1" = build ]; then
  python -m $P.build >/dev/null
  for k in ${VARIANTS:-1 2 3 4 32}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTT_CE16_EXP=$k -Iinclude -I$P/csrc -x hip -c $P/csrc/ce_f16x2.hip -o $P/lib/exp/ce16_e$k.o &
  done
  wait
  for k in ${VARIANTS:-1 2 3 4 32}; do
    objs=$(ls $P/csrc/_obj/*.o | grep -v "/ce_f16x2.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/exp/libtt_hotpath_c$k.so $objs $P/lib/exp/ce16_e$k.o
  done
  ls $P/lib/exp/*_c*.so
else
  for rep in 1 2; do
    python tools/bench_ce.py 8192 65536 2>&1 | grep f16x2 | grep -v pair | sed -e 's/^/product : /'
    for f in $P/lib/exp/libtt_hotpath_c*.so; do
      TT_HOTPATH_LIB=$PWD/$f python tools/bench_ce.py 8192 65536 2>&1 | grep f16x2 | grep -v pair | sed -e "s#^#$(basename $f .so | sed s/libtt_hotpath_//) : #"
    done
  done
fi
