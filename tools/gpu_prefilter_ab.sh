#!/bin/bash
# A/B of the hits-mode fp32 prefilter (k_letters_pre) on the sequence-only workload C2, plus the all-scores line.
set -e
mkdir -p gpurun_out
: > gpurun_out/prefilter_ab.log
echo "== c2 all-scores" >> gpurun_out/prefilter_ab.log
python3 bench.py --workload c2 --no-cpu-baseline --steps 30 2>>gpurun_out/prefilter_ab.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
" >> gpurun_out/prefilter_ab.log
for m in 30 12 8 6 4 0 -4; do
  for pf in 1 0; do
    echo "== minscore $m PFMSCAN_PREFILTER=$pf" >> gpurun_out/prefilter_ab.log
    PFMSCAN_PREFILTER=$pf python3 bench.py --workload c2 --mode hits --minscore " $m" --no-cpu-baseline --steps 30 2>>gpurun_out/prefilter_ab.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f hits %s' % (d['ms_per_step'], d['config'].get('hits_per_step')))
" >> gpurun_out/prefilter_ab.log
  done
done
