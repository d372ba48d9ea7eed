#!/bin/bash
# the kernel traces of the two all-scores headlines again (bench.py no longer times default-allocator arrays under a profiler)
ulimit -c 0
R=r5; OUT=gpurun_out/$R; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
export PFMSCAN_BENCH_NO_FLOOR=1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c3 -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 200 > $ROOT/$OUT/trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_c2 -- python3 $ROOT/bench.py --workload c2 --width 8 --no-cpu-baseline --steps 200 > $ROOT/$OUT/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace_default -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 > $ROOT/$OUT/trace_default.log 2>&1
cd $ROOT
for t in trace_c3 trace_c2 trace_default; do f=$(ls -t $(find $OUT/$t -name "*kernel_stats.csv") | head -1); [ -n "$f" ] && cp $f $OUT/${t}_kernel_stats.csv; grep "^{\"metric\"" $OUT/$t.log | tail -1 > $OUT/${t}_bench_line.json; done
head -3 $OUT/trace_c3_kernel_stats.csv | cut -c1-170; head -3 $OUT/trace_c2_kernel_stats.csv | cut -c1-170
python3 -c "
import json
for t in ('trace_c3','trace_c2'):
    d=json.load(open('$OUT/'+t+'_bench_line.json')); print(t, d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step'])"
