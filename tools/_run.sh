cd $GRAFT_REPO_ROOT/_r2
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3fuzz; mkdir -p $OUT
GRAFT_REPO_ROOT=$GRAFT_REPO_ROOT/_r2 timeout 300 python tools/fuzz_sharded.py 150 32 > $OUT/fuzz_sharded_r2tree.txt 2>&1; grep "FINDING\|cases," $OUT/fuzz_sharded_r2tree.txt
