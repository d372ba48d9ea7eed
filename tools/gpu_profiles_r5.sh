#!/bin/bash
# Round-5 evidence in ONE gpurun call (same box for all lines): bench lines, kernel traces, PMC passes, CLI end to end.
# usage: tools/gpu_profiles_r5.sh   -> gpurun_out/r4/ ; copy what is to be kept into profiles/r5/
R=r5
OUT=gpurun_out/$R
mkdir -p $OUT
ulimit -c 0
ROOT=$(pwd)
run() { name=$1; shift; echo "== $name: bench.py $*"; if [ "$name" = bench_c3_generic_kernel ]; then export PFMSCAN_PROFILE_GENERIC=1; else unset PFMSCAN_PROFILE_GENERIC; fi; python3 bench.py "$@" 2>>$OUT/err.log | tail -1 > $OUT/$name.json; python3 -c "
import json,sys; d=json.load(open('$OUT/$name.json')); print('   ms_per_step %.4f value %.4g %s n_gpus %d' % (d['ms_per_step'], d['value'], d['unit'], d['n_gpus']))"; }
run bench_c3_default
run bench_c5_library --workload c5 --steps 10 --warmup 2 --no-cpu-baseline
run bench_c5_library_f64 --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --profile-dtype float64
run bench_c5s_struct_library --workload c5s --steps 3 --warmup 1 --no-cpu-baseline
run bench_c3_profile_f64 --no-cpu-baseline --no-secondary --profile-dtype float64
run bench_c3_hits --mode hits --no-cpu-baseline
run bench_c3_hits_two_phase --mode hits2 --no-cpu-baseline
run bench_c2_seq_only --workload c2 --width 8 --no-cpu-baseline --steps 50
PFMSCAN_LETTERS_GENERIC=1 python3 bench.py --workload c2 --width 8 --no-cpu-baseline --steps 50 2>>$OUT/err.log | tail -1 > $OUT/bench_c2_generic_kernel.json
tools/hbm_mixed 100000 3000 c2 > $OUT/hbm_mixed_c2.txt 2>&1
tools/hbm_mixed 100000 3000 c2 placed > $OUT/hbm_mixed_c2_placed.txt 2>&1
run bench_c2_hits_m6 --workload c2 --width 8 --mode hits --minscore-seq 6 --no-cpu-baseline
run bench_c2_hits_none --workload c2 --width 8 --mode hits --minscore-seq 30 --no-cpu-baseline
run bench_c2_w4_hits_m2 --workload c2 --width 4 --mode hits --minscore-seq 2 --no-cpu-baseline
run bench_c4_shard_125k --records 125000 --no-cpu-baseline --no-secondary
tools/hbm_mixed > $OUT/hbm_mixed_ceiling.txt 2>&1
tools/hbm_mixed 100000 3000 placed > $OUT/hbm_mixed_ceiling_placed.txt 2>&1
run bench_c3_torch_allocator --no-cpu-baseline --no-secondary --placement torch
run bench_c3_generic_kernel --no-cpu-baseline --no-secondary
unset PFMSCAN_PROFILE_GENERIC
cd /tmp && export TMPDIR=/tmp
export PFMSCAN_BENCH_NO_FLOOR=1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c3 -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 200 > $ROOT/$OUT/trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c5 -- python3 $ROOT/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/trace_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c2 -- python3 $ROOT/bench.py --workload c2 --width 8 --no-cpu-baseline --steps 200 > $ROOT/$OUT/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c2_hits -- python3 $ROOT/bench.py --workload c2 --width 8 --mode hits --minscore-seq 6 --no-cpu-baseline --steps 200 > $ROOT/$OUT/trace_c2_hits.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_default -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 > $ROOT/$OUT/trace_default.log 2>&1
cd $ROOT
for t in trace_c3 trace_c5 trace_c2 trace_c2_hits trace_default; do f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv; done
# PMC passes (counters only, each pass its own process)
BENCH_ARGS="--no-secondary" tools/pmc.sh r5_c3 > $OUT/pmc_c3.log 2>&1
cp gpurun_out/pmc_r5_c3/summary.txt $OUT/bench_c3_pmc_summary.txt 2>/dev/null
BENCH_ARGS="--workload c2 --width 8" tools/pmc.sh r5_c2 > $OUT/pmc_c2.log 2>&1
cp gpurun_out/pmc_r5_c2/summary.txt $OUT/bench_c2_pmc_summary.txt 2>/dev/null
BENCH_ARGS="--workload c2 --width 8 --mode hits --minscore-seq 6" tools/pmc.sh r5_c2hits > $OUT/pmc_c2hits.log 2>&1
cp gpurun_out/pmc_r5_c2hits/summary.txt $OUT/bench_c2_hits_pmc_summary.txt 2>/dev/null
BENCH_ARGS="--workload c5 --steps 2 --warmup 1" tools/pmc.sh r5_c5 > $OUT/pmc_c5.log 2>&1
cp gpurun_out/pmc_r5_c5/summary.txt $OUT/bench_c5_pmc_summary.txt 2>/dev/null
# the default line's secondary legs (SS hits: k_letters_cred8; two-FASTA: letters pass + k_letters_at) under the counters
BENCH_ARGS="--steps 3" tools/pmc.sh r5_default > $OUT/pmc_default.log 2>&1
cp gpurun_out/pmc_r5_default/summary.txt $OUT/bench_default_pmc_summary.txt 2>/dev/null
# the command line end to end: 20k x 3 kb and 100k x 3 kb, all four modes of the README (RNA, SS, RNASS with a store, RNASS two FASTA)
CLI_E2E_LIBRARY=256 python3 tools/cli_e2e.py 20000 3000 100000 store > $OUT/cli_end_to_end.txt 2> $OUT/cli_end_to_end.err
ls $OUT
