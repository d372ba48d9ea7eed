#!/bin/bash
# A/B of the integer position-keyed prefilter (k_letters_cred) against the fp32 one (k_letters_pre), same box, same call.
# usage: tools/gpu_ab_credits.sh <outdir>
OUT=${1:-gpurun_out/ab_credits}
mkdir -p $OUT
for cfg in "c2 8 6" "c2 8 30" "c2 12 6" "c2 16 6" "c2 4 2"; do
  set -- $cfg
  for cr in 1 0; do
    PFMSCAN_CREDITS=$cr python3 bench.py --workload $1 --width $2 --mode hits --minscore-seq $3 --no-cpu-baseline --steps 50 2>/dev/null | tail -1 > $OUT/ab_$1_w$2_m$3_cred$cr.json
    python3 -c "
import json; d=json.load(open('$OUT/ab_$1_w$2_m$3_cred$cr.json')); print('w=$2 thr=$3 credits=$cr  kernel_ms median %.4f min %.4f  hits %d' % (d['roofline']['kernel_ms_median'], d['roofline']['kernel_ms_min'], d['config']['hits_per_step']))"
  done
done
# combined two-phase (C3 hits2) with and without
for cr in 1 0; do
  PFMSCAN_CREDITS=$cr python3 bench.py --mode hits2 --no-cpu-baseline --steps 50 2>/dev/null | tail -1 > $OUT/ab_c3_hits2_cred$cr.json
  python3 -c "
import json; d=json.load(open('$OUT/ab_c3_hits2_cred$cr.json')); print('c3 hits2 credits=$cr  ms_per_step %.4f hits %d' % (d['ms_per_step'], d['config']['hits_per_step']))"
done
