#!/bin/bash
# A/B of environment knobs with ONE build in ONE gpurun call:
#   tools/gpu_ab_env.sh "KNOB=0 KNOB=1" <bench.py args>   -> gpurun_out/ab_env.log
set -e
VARS=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  for V in $VARS; do
    echo "== $V $*" >> gpurun_out/ab_env.log
    env $V python3 bench.py "$@" --no-cpu-baseline 2>>gpurun_out/ab_env.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f hits %s' % (d['ms_per_step'], d['config'].get('hits_per_step')))
" >> gpurun_out/ab_env.log
  done
done
