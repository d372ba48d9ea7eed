#!/bin/bash
# C2 (BASELINE config 2: sequence-only all-scores, w = 8): parity of the fixed-width kernel, A/B against the width-generic one on one
# box, the floor of C2's byte mix, widths 4..16, and fresh PMC passes.   -> gpurun_out/r5_c2/
ulimit -c 0
O=gpurun_out/r5_c2; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_wide.py tests/test_gpu_property.py -x -q -k "pwm or letters or seq or c2 or stream or wide or property" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
line() { python3 -c "
import json,sys; d=json.load(open('$1')); r=d['roofline']; f=r.get('mixed_read_write_floor') or {}
print('$2', 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'min', round(r['kernel_ms_min'],4), 'frac', round(r['frac'],3), 'default_alloc', r.get('kernel_ms_default_allocator'), 'floor', f.get('ms'), r['kernel'])"; }
for r in 1 2; do
  for v in fixed generic; do
    if [ $v = generic ]; then export PFMSCAN_LETTERS_GENERIC=1; else unset PFMSCAN_LETTERS_GENERIC; fi
    python3 bench.py --workload c2 --width 8 --no-cpu-baseline --steps 50 2>>$O/err.log | tail -1 > $O/bench_c2_$v.json; line $O/bench_c2_$v.json "c2 w8 $v"
  done
done
unset PFMSCAN_LETTERS_GENERIC
python3 bench.py --workload c2 --width 8 --no-cpu-baseline --steps 50 --placement torch 2>>$O/err.log | tail -1 > $O/bench_c2_torch.json; line $O/bench_c2_torch.json "c2 w8 fixed torch-alloc"
for w in 4 6 12 16; do
  for v in fixed generic; do
    if [ $v = generic ]; then export PFMSCAN_LETTERS_GENERIC=1; else unset PFMSCAN_LETTERS_GENERIC; fi
    PFMSCAN_BENCH_NO_FLOOR=1 python3 bench.py --workload c2 --width $w --no-cpu-baseline --steps 50 2>>$O/err.log | tail -1 > $O/bench_c2_w${w}_$v.json; line $O/bench_c2_w${w}_$v.json "c2 w$w $v"
  done
done
unset PFMSCAN_LETTERS_GENERIC
tools/hbm_mixed 100000 3000 c2 > $O/hbm_mixed_c2.txt 2>&1; tail -12 $O/hbm_mixed_c2.txt
tools/hbm_mixed 100000 3000 c2 placed > $O/hbm_mixed_c2_placed.txt 2>&1; tail -3 $O/hbm_mixed_c2_placed.txt
BENCH_ARGS="--workload c2 --width 8" tools/pmc.sh r5_c2 > $O/pmc_c2.log 2>&1
cp gpurun_out/pmc_r5_c2/summary.txt $O/bench_c2_pmc_summary.txt 2>/dev/null; tail -30 $O/bench_c2_pmc_summary.txt
