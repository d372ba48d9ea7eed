#!/bin/bash
# A/B of several builds of libpfmscan in ONE gpurun call (boxes differ by several % between calls):
#   tools/gpu_ab_lib.sh "<libA.so> <libB.so> ..." <bench.py args...>   -> gpurun_out/ab_lib.log
set -e
LIBS=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  for L in $LIBS; do
    echo "== $L $*" >> gpurun_out/ab_lib.log
    PFMSCAN_LIB=$PWD/$L python3 bench.py "$@" --no-cpu-baseline 2>>gpurun_out/ab_lib.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f' % d['ms_per_step'])
" >> gpurun_out/ab_lib.log
  done
done
