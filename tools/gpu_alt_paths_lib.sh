#!/bin/bash
# the library tests once per alternate library path (see tools/gpu_alt_paths.sh)
ulimit -c 0
mkdir -p gpurun_out/r5
T="tests/test_gpu_library.py tests/test_gpu_library8.py tests/test_gpu_proflib.py tests/test_gpu_threshold_exact.py tests/test_gpu_fuzz.py"
for knob in PFMSCAN_LIB_SORT=1 PFMSCAN_LIB_SEQUENTIAL=1 PFMSCAN_FORCE_GENERIC=1 PFMSCAN_CRED8_NJ=16 PFMSCAN_PROFILE_FIXED_MIN=4 PFMSCAN_UPLOAD_THREADS=1; do
  echo "== $knob"
  env $knob timeout -k 10 700 python3 -m pytest $T -x -q -o faulthandler_timeout=400 2>&1 | tail -3 || exit 1
done
