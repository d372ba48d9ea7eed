#!/bin/bash
# A/B of the single-motif hits kernels on one box, one call:
#   r2    round-2 build: k_letters_cred, pair index by shift / or / mask
#   cred  this build, PFMSCAN_QUAD=0: k_letters_cred, pair index by v_dot4_u32_u8
#   quad  this build, PFMSCAN_QUAD=1: k_letters_quad (four-letter credit tables)
# usage: tools/gpu_ab_quad.sh [outdir]
OUT=${1:-gpurun_out/r3_ab_quad}
mkdir -p $OUT
: > $OUT/ab.txt
run() { # label, env..., -- bench args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python3 bench.py "$@" --no-cpu-baseline --no-secondary --steps 50 2>>$OUT/err.log | tail -1 > $OUT/tmp.json
  python3 -c "
import json; d=json.load(open('$OUT/tmp.json')); r=d['roofline'] or {}
print('%-5s %-44s ms_per_step %.4f (median %s min %s) hits %s' % ('$label', '$*', d['ms_per_step'], r.get('kernel_ms_median'), r.get('kernel_ms_min'), d['config']['hits_per_step']))" | tee -a $OUT/ab.txt
}
R2=$(pwd)/rnascan_amd/libpfmscan_r2.so
for round in 1 2; do
for cfg in "8 6" "8 30" "12 6" "16 6" "4 2" "18 6" "24 6" "32 6"; do
  set -- $cfg
  [ -f $R2 ] && run r2 PFMSCAN_LIB=$R2 -- --workload c2 --width $1 --mode hits --minscore-seq $2
  run cred PFMSCAN_QUAD=0 -- --workload c2 --width $1 --mode hits --minscore-seq $2
  run quad PFMSCAN_QUAD=1 -- --workload c2 --width $1 --mode hits --minscore-seq $2
done
[ -f $R2 ] && run r2 PFMSCAN_LIB=$R2 -- --mode hits2
run cred PFMSCAN_QUAD=0 -- --mode hits2
run quad PFMSCAN_QUAD=1 -- --mode hits2
done
