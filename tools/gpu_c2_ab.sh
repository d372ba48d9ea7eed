#!/bin/bash
# C2 A/B on one box: fixed-width vs width-generic letters kernel at several widths -> gpurun_out/r5_c2/
ulimit -c 0
O=gpurun_out/r5_c2; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_wide.py tests/test_gpu_property.py -x -q -k "pwm or letters or seq or c2 or stream or wide or property" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
line() { python3 -c "
import json,sys; d=json.load(open('$1')); r=d['roofline']; f=r.get('mixed_read_write_floor') or {}
print('$2', 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'min', round(r['kernel_ms_min'],4), 'frac', round(r['frac'],3), 'default_alloc', r.get('kernel_ms_default_allocator'), 'floor', f.get('ms'), r['kernel'])"; }
for w in ${WIDTHS:-8 8 4 6 10 12 16 20 40}; do
  for v in fixed generic; do
    if [ $v = generic ]; then export PFMSCAN_LETTERS_GENERIC=1; else unset PFMSCAN_LETTERS_GENERIC; fi
    PFMSCAN_BENCH_NO_FLOOR=${NOFLOOR:-1} python3 bench.py --workload c2 --width $w --no-cpu-baseline --steps 50 2>>$O/err.log | tail -1 > $O/bench_c2_w${w}_$v.json; line $O/bench_c2_w${w}_$v.json "c2 w$w $v"
  done
done
