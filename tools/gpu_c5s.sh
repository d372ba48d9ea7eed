#!/bin/bash
# struct-only library bench lines (k_profile_lib): tools/gpu_c5s.sh [extra bench args]  -> gpurun_out/r3b/
mkdir -p gpurun_out/r3b
for v in "" "--profile-dtype float64" "--variant inf"; do
  tag=c5s$(echo $v | tr -d ' -')
  python3 bench.py --workload c5s --steps 3 --warmup 1 --no-cpu-baseline $v "$@" 2>>gpurun_out/r3b/err.log | tail -1 > gpurun_out/r3b/$tag.json
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r3b/$tag.json")); print("$tag", round(d["ms_per_step"],2), "ms  frac", round(d["roofline"]["frac"],3), "hits", d["config"]["hits_per_step"], "thr", d["config"]["minscore_struct"])
PY
done
