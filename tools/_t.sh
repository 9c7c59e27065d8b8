#!/bin/bash
# scratch GPU job (not committed)
cd /root/repo
python -m pytest tests -m gpu -q -x -p no:warnings 2>&1 | grep -v "^$" | tail -15 > gpurun_out/t_pytest.txt
for wl in P C2 C3; do python tools/ab_c3.py $wl optim._MARKED 200 2 2>&1 | grep "_MARKED" ; done > gpurun_out/t_ab.txt 2>&1
for b in 2 4; do TT_SWEEP_BATCH=$b python tools/ab_c3.py P optim._MARKED 200 1 2>&1 | grep "_MARKED=True" | sed "s/^/batch$b /"; done >> gpurun_out/t_ab.txt 2>&1
cat gpurun_out/t_pytest.txt gpurun_out/t_ab.txt
