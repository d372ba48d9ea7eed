#!/bin/bash
# Round-2 evidence in ONE gpurun call (same box for all lines): bench lines, kernel traces, PMC passes.
# usage: tools/gpu_profiles_r2.sh   -> gpurun_out/r2/ ; copy what is to be kept into profiles/r2/
R=r2
OUT=gpurun_out/$R
mkdir -p $OUT
ROOT=$(pwd)
run() { name=$1; shift; echo "== $name: bench.py $*"; python3 bench.py "$@" 2>>$OUT/err.log | tail -1 > $OUT/$name.json; python3 -c "
import json,sys; d=json.load(open('$OUT/$name.json')); print('   ms_per_step %.4f value %.4g %s' % (d['ms_per_step'], d['value'], d['unit']))"; }
run bench_c3_default
run bench_c5_library --workload c5 --steps 10 --warmup 2 --no-cpu-baseline
run bench_c5_library_inf --workload c5 --variant inf --steps 5 --warmup 1 --no-cpu-baseline
run bench_c3_hits --mode hits --no-cpu-baseline
run bench_c3_hits_two_phase --mode hits2 --no-cpu-baseline
run bench_c2_seq_only --workload c2 --width 8 --no-cpu-baseline
run bench_c2_hits_m6 --workload c2 --width 8 --mode hits --minscore-seq 6 --no-cpu-baseline
run bench_c2_hits_none --workload c2 --width 8 --mode hits --minscore-seq 30 --no-cpu-baseline
run bench_c3_from_host --from-host --steps 3 --warmup 1 --settle 1 --no-cpu-baseline
run bench_c4_shard_125k --records 125000 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
# kernel traces (per-kernel stats over the same commands; 200 timed steps so that the ramp does not dominate the average)
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c3 -- python3 $ROOT/bench.py --no-cpu-baseline --steps 200 > $ROOT/$OUT/trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c5 -- python3 $ROOT/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/trace_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c2_hits -- python3 $ROOT/bench.py --workload c2 --width 8 --mode hits --minscore-seq 6 --no-cpu-baseline --steps 200 > $ROOT/$OUT/trace_c2_hits.log 2>&1
cd $ROOT
for t in trace_c3 trace_c5 trace_c2_hits; do f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv; done
# PMC passes
BENCH_ARGS="--workload c5 --steps 2 --warmup 1" tools/pmc.sh r2_c5 > $OUT/pmc_c5.log 2>&1
cp gpurun_out/pmc_r2_c5/summary.txt $OUT/bench_c5_pmc_summary.txt 2>/dev/null
BENCH_ARGS="--workload c2 --width 8 --mode hits --minscore-seq 6" tools/pmc.sh r2_c2hits > $OUT/pmc_c2hits.log 2>&1
cp gpurun_out/pmc_r2_c2hits/summary.txt $OUT/bench_c2_hits_pmc_summary.txt 2>/dev/null
tools/pmc.sh r2_c3 > $OUT/pmc_c3.log 2>&1
cp gpurun_out/pmc_r2_c3/summary.txt $OUT/bench_c3_pmc_summary.txt 2>/dev/null
ls $OUT
