#!/bin/bash
set -e
: > gpurun_out/jch.log
for j in 1 2 4; do
  for wd in 8 12 24; do
  echo "== JCH $j width $wd" >> gpurun_out/jch.log
  PFMSCAN_LIB=$PWD/rnascan_amd/libpfmscan_j$j.so python3 bench.py --workload c2 --width $wd --no-cpu-baseline --steps 30 2>>gpurun_out/jch.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f' % d['ms_per_step'])
" >> gpurun_out/jch.log
  done
done
