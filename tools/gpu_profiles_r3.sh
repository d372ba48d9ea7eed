#!/bin/bash
# Round-3 evidence in ONE gpurun call (same box for all lines): bench lines, kernel traces, PMC passes.
# usage: tools/gpu_profiles_r3.sh   -> gpurun_out/r3/ ; copy what is to be kept into profiles/r3/
R=r3
OUT=gpurun_out/$R
mkdir -p $OUT
ROOT=$(pwd)
run() { name=$1; shift; echo "== $name: bench.py $*"; python3 bench.py "$@" 2>>$OUT/err.log | tail -1 > $OUT/$name.json; python3 -c "
import json,sys; d=json.load(open('$OUT/$name.json')); print('   ms_per_step %.4f value %.4g %s n_gpus %d' % (d['ms_per_step'], d['value'], d['unit'], d['n_gpus']))"; }
run bench_c3_default
run bench_c5_library --workload c5 --steps 10 --warmup 2 --no-cpu-baseline
run bench_c5_library_f64 --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --profile-dtype float64
run bench_c5_library_inf --workload c5 --variant inf --steps 5 --warmup 1 --no-cpu-baseline
run bench_c5s_struct_library --workload c5s --steps 3 --warmup 1 --no-cpu-baseline
run bench_c5s_struct_library_f64 --workload c5s --steps 3 --warmup 1 --no-cpu-baseline --profile-dtype float64
run bench_c5s_struct_library_inf --workload c5s --variant inf --steps 3 --warmup 1 --no-cpu-baseline
run bench_c3_profile_f64 --no-cpu-baseline --no-secondary --profile-dtype float64
run bench_c3_hits --mode hits --no-cpu-baseline
run bench_c3_hits_two_phase --mode hits2 --no-cpu-baseline
run bench_c2_seq_only --workload c2 --width 8 --no-cpu-baseline
run bench_c2_hits_m6 --workload c2 --width 8 --mode hits --minscore-seq 6 --no-cpu-baseline
run bench_c2_hits_none --workload c2 --width 8 --mode hits --minscore-seq 30 --no-cpu-baseline
run bench_c2_w4_hits_m2 --workload c2 --width 4 --mode hits --minscore-seq 2 --no-cpu-baseline
run bench_c3_from_host --from-host --steps 3 --warmup 1 --settle 1 --no-cpu-baseline
run bench_c4_shard_125k --records 125000 --no-cpu-baseline --no-secondary
# N GPUs from one command, rehearsed on the one GPU of the box: two ranks share device 0, gloo rendezvous
echo "== bench_c3_gpus2_rehearsal"
PFMSCAN_BENCH_REHEARSE=1 python3 bench.py --gpus 2 --steps 20 --warmup 3 --no-ref-structured 2>>$OUT/err.log | tail -1 > $OUT/bench_c3_gpus2_rehearsal_one_device.json
python3 -c "
import json; d=json.load(open('$OUT/bench_c3_gpus2_rehearsal_one_device.json')); print('   n_gpus', d['n_gpus'], 'per_rank', d['per_rank'])"
# the RCCL process group with the one rank a one-GPU box allows
echo "== bench_c3_rccl_one_rank"
PFMSCAN_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 1 --steps 20 --warmup 3 --no-ref-structured 2>>$OUT/err.log | tail -1 > $OUT/bench_c3_rccl_one_rank.json
cd /tmp && export TMPDIR=/tmp
# kernel traces (per-kernel stats over the same commands; 200 timed steps so that the ramp does not dominate the average)
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c3 -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 200 > $ROOT/$OUT/trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c5 -- python3 $ROOT/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/trace_c5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c5_f64 -- python3 $ROOT/bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --profile-dtype float64 > $ROOT/$OUT/trace_c5_f64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c5s -- python3 $ROOT/bench.py --workload c5s --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/trace_c5s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c2_hits -- python3 $ROOT/bench.py --workload c2 --width 8 --mode hits --minscore-seq 6 --no-cpu-baseline --steps 200 > $ROOT/$OUT/trace_c2_hits.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_default -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 > $ROOT/$OUT/trace_default.log 2>&1
cd $ROOT
for t in trace_c3 trace_c5 trace_c5_f64 trace_c5s trace_c2_hits trace_default; do f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv; done
# PMC passes
BENCH_ARGS="--workload c5 --steps 2 --warmup 1" tools/pmc.sh r3_c5 > $OUT/pmc_c5.log 2>&1
cp gpurun_out/pmc_r3_c5/summary.txt $OUT/bench_c5_pmc_summary.txt 2>/dev/null
BENCH_ARGS="--workload c5s --steps 2 --warmup 1" tools/pmc.sh r3_c5s > $OUT/pmc_c5s.log 2>&1
cp gpurun_out/pmc_r3_c5s/summary.txt $OUT/bench_c5s_pmc_summary.txt 2>/dev/null
BENCH_ARGS="--workload c2 --width 8 --mode hits --minscore-seq 6" tools/pmc.sh r3_c2hits > $OUT/pmc_c2hits.log 2>&1
cp gpurun_out/pmc_r3_c2hits/summary.txt $OUT/bench_c2_hits_pmc_summary.txt 2>/dev/null
BENCH_ARGS="--no-secondary" tools/pmc.sh r3_c3 > $OUT/pmc_c3.log 2>&1
cp gpurun_out/pmc_r3_c3/summary.txt $OUT/bench_c3_pmc_summary.txt 2>/dev/null
ls $OUT
