#!/bin/bash
# PMC passes for the bench kernel (each pass = its own process, counters only).
# extra bench.py arguments: BENCH_ARGS="--workload c2 --width 8"
# usage: tools/pmc.sh <tag> ["ENV=.. ENV=.."] ; output: gpurun_out/pmc_<tag>/pass*/ + summary
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-default}
VARIANT=${2:-}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PFMSCAN_BENCH_NO_FLOOR=1      # bench.py starts no child process (its live floor measurement) under the profiler
for kv in $VARIANT; do export $kv; done
i=0
for counters in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $counters --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; }
done
python3 $ROOT/tools/pmc_summary.py $OUT | tee $OUT/summary.txt
