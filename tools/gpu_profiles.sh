#!/bin/bash
# Regenerate the bench lines kept under profiles/<round>/ in ONE gpurun call (same box for all of them).
# usage: tools/gpu_profiles.sh <round>      -> gpurun_out/<round>/*.json ; copy what is to be kept into profiles/<round>/
set -e
R=${1:-r1}
OUT=gpurun_out/$R
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: bench.py $*"; python3 bench.py "$@" 2>>$OUT/err.log | tail -1 > $OUT/$name.json; python3 -c "
import json,sys; d=json.load(open('$OUT/$name.json')); print('   ms_per_step %.4f value %.4g %s' % (d['ms_per_step'], d['value'], d['unit']))"; }
run bench_c3_default
run bench_c3_variant_inf --variant inf --no-ref-structured --no-cpu-baseline
run bench_c3_profile_f64 --profile-dtype float64 --no-ref-structured
run bench_c3_hits --mode hits --no-cpu-baseline
run bench_c3_hits_two_phase --mode hits2 --no-cpu-baseline
run bench_c3_hits_two_phase_m2 --mode hits2 --minscore 2 --no-cpu-baseline
run bench_c2_seq_only --workload c2 --width 8 --no-cpu-baseline
run bench_c2_hits_m6 --workload c2 --width 8 --mode hits --minscore 6 --no-cpu-baseline
run bench_c2_hits_none --workload c2 --width 8 --mode hits --minscore 30 --no-cpu-baseline
run bench_c5_256_motifs --workload c5 --motifs 256 --steps 3 --warmup 1 --no-cpu-baseline
