#!/bin/bash
# C5: the split form (prefilter launch + verify launch) against the fused kernel on one box -> gpurun_out/r5_c5/
ulimit -c 0
O=gpurun_out/r5_c5; mkdir -p $O
PFMSCAN_LIB_SPLIT=1 timeout -k 5 600 python -m pytest tests/test_gpu_library.py tests/test_gpu_threshold_exact.py -x -q -k "library" > $O/pytest_split.txt 2>&1; tail -3 $O/pytest_split.txt
line() { python3 -c "
import json; d=json.load(open('$1')); print('$2', round(d['ms_per_step'],3), d['config']['hits_per_step'])"; }
for r in 1 2; do
  for v in ${VARIANTS:-split fused}; do
    if [ $v = fused ]; then export PFMSCAN_LIB_SPLIT=0; else unset PFMSCAN_LIB_SPLIT; fi
    python3 bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline 2>>$O/err.log | tail -1 > $O/c5_$v.json; line $O/c5_$v.json "c5 $v"
    python3 bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline --profile-dtype float64 2>>$O/err.log | tail -1 > $O/c5_f64_$v.json; line $O/c5_f64_$v.json "c5 f64 $v"
  done
done
unset PFMSCAN_LIB_SPLIT
