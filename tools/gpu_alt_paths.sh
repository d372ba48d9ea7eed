#!/bin/bash
# the parity / fuzz / big-stream / threshold tests once per alternate code path (environment knobs of INTEGRATION.md): kernels that
# the default dispatch does not pick must give the same results.  One pytest process per knob, one after the other.
ulimit -c 0
mkdir -p gpurun_out/r5
T="tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_threshold_exact.py tests/test_gpu_big_streams.py tests/test_gpu_letters8.py tests/test_gpu_property.py"
for knob in PFMSCAN_QUAD=1 PFMSCAN_TWO_PHASE=0 PFMSCAN_PROFILE_GENERIC=1 PFMSCAN_LETTERS_GENERIC=1 PFMSCAN_CREDITS=0 PFMSCAN_PREFILTER=0 PFMSCAN_DMA=0 PFMSCAN_V=3 PFMSCAN_FORCE_GENERIC=1 PFMSCAN_PAIR_TWO_PHASE=1 PFMSCAN_TILES_PER_BLOCK=1; do
  echo "== $knob"
  env $knob timeout -k 10 500 python3 -m pytest $T -x -q -o faulthandler_timeout=300 2>&1 | tail -3 || exit 1
done
