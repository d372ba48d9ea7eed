#!/bin/bash
# same as gpu_sweep.sh for the sequence-only workload (C2)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/sweep_c2.log
for variant in "$@"; do
  echo "== $variant" >> $OUT/sweep_c2.log
  env $variant python3 $ROOT/bench.py --workload c2 --width 8 --steps 20 --warmup 3 --no-cpu-baseline 2>>$OUT/sweep.err | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.4f kernel_ms %.4f frac %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac']))" >> $OUT/sweep_c2.log
done
cat $OUT/sweep_c2.log
