#!/bin/bash
set -e
: > gpurun_out/pre_ablate.log
for ab in 0 1 2 3 8 9 10 11; do
  echo "== ablate $ab" >> gpurun_out/pre_ablate.log
  PFMSCAN_ABLATE=$ab python3 bench.py --workload c2 --mode hits --minscore 30 --no-cpu-baseline --steps 30 2>>gpurun_out/pre_ablate.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f' % d['ms_per_step'])
" >> gpurun_out/pre_ablate.log
done
