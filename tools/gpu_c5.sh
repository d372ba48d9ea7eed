#!/bin/bash
# C5 lines of the library kernel on one box: float32 and float64 profile rows, optionally several builds
# usage: tools/gpu_c5.sh [lib.so ...]   (default: the in-tree build)
mkdir -p gpurun_out/r3c5
LIBS=${@:-libpfmscan.so}
for round in 1 2; do
for lib in $LIBS; do
  for v in "" "--profile-dtype float64"; do
    PFMSCAN_LIB=$(pwd)/rnascan_amd/$lib python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline $v 2>>gpurun_out/r3c5/err.log | tail -1 > gpurun_out/r3c5/tmp.json
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r3c5/tmp.json")); print("$lib", "$v", round(d["ms_per_step"],3), "ms  hits", d["config"]["hits_per_step"], "lds frac", round(d["roofline"]["frac"],3))
PY
  done
done
done
