#!/bin/bash
# Build measurement variants of the MIPS pass-1 kernel (mips.hip -DTT_MIPS_EXP=k) next to the product library and time
# each with tools/bench_mips.py: which part of a tile's time goes where.  Variant results are WRONG by design.
#   bit 1 (2): one LDS read per sub-tile   bit 2 (4): no per-tile barrier   bit 3 (8): no result stores
#   bit 4 (16): corpus tiles from a 64-chunk window (L2 hits)   bit 5 (32): result stores into an 8-chunk window (L2)
#   bit 6 (64): bare s_barrier instead of __syncthreads() (no vmcnt(0) drain)   +128: one tile fewer in flight   +256: vmcnt(0) + bare barrier
#   "row": the row-exact epilogue (4 VALU / score) instead of the quad one -- this one is correct, just slower
#   tools/mips_variants.sh build      (here: hipcc cross-compiles)        tools/mips_variants.sh run   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
P=two_tower_models_amd
mkdir -p $P/lib/exp
if [ "$1" = build ]; then
  python -m $P.build >/dev/null
  for k in ${VARIANTS:-2 4 8 16 24 32 row}; do
    if [ $k = row ]; then defs="-DTT_MIPS_ROWARG"; else defs="-DTT_MIPS_EXP=$k"; fi
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $defs -Iinclude -I$P/csrc -x hip -c $P/csrc/mips.hip -o $P/lib/exp/mips_e$k.o &
  done
  wait
  for k in ${VARIANTS:-2 4 8 16 24 32 row}; do
    objs=$(ls $P/csrc/_obj/*.o | grep -v "/mips.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/lib/exp/libtt_hotpath_e$k.so $objs $P/lib/exp/mips_e$k.o
  done
  ls $P/lib/exp/*.so
else
  for rep in 1 2; do
    python tools/bench_mips.py 2>&1 | grep bf16 | sed -e 's/^bf16: C=[0-9]* B=[0-9]* K=[0-9]*: //' -e 's/, corpus stream.*//' -e 's/^/product : /'
    for f in $P/lib/exp/libtt_hotpath_e*.so; do
      TT_HOTPATH_LIB=$PWD/$f python tools/bench_mips.py 2>&1 | grep bf16 | sed -e 's/^bf16: C=[0-9]* B=[0-9]* K=[0-9]*: //' -e 's/, corpus stream.*//' -e "s#^#$(basename $f .so | sed s/libtt_hotpath_//) : #"
    done
  done
fi
