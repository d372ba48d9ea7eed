#!/bin/bash
# A/B sweep of the k_profile tuning knobs on the GPU box (one process per variant).
# usage: tools/gpu_sweep.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one variant's env)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/sweep.log
for variant in "$@"; do
  echo "== $variant" >> $OUT/sweep.log
  env $variant python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>>$OUT/sweep.err | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))" >> $OUT/sweep.log
done
cat $OUT/sweep.log
