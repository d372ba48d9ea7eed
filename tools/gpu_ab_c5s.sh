#!/bin/bash
# A/B of k_profile_lib builds on one box: tools/gpu_ab_c5s.sh lib1.so lib2.so ...
mkdir -p gpurun_out/r3b
for round in 1 2; do
for lib in "$@"; do
  for v in "" "--variant inf"; do
    PFMSCAN_LIB=$(pwd)/rnascan_amd/$lib python3 bench.py --workload c5s --steps 3 --warmup 1 --no-cpu-baseline $v 2>>gpurun_out/r3b/err.log | tail -1 > gpurun_out/r3b/ab.json
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r3b/ab.json")); print("$lib", "$v", round(d["ms_per_step"],2), "ms  frac", round(d["roofline"]["frac"],3), "hits", d["config"]["hits_per_step"])
PY
  done
done
done
