#!/bin/bash
# tiles per workgroup of the single-motif hits kernels (PFMSCAN_TILES_PER_BLOCK) on C2: the grid against the resident capacity
# usage: tools/gpu_sweep_tpb.sh ["ENV=.. ENV=.."] [bench args]
mkdir -p gpurun_out/r3s
for kv in $1; do export $kv; done
shift
for round in 1 2; do
for tpb in 0 6 8 12 16 24 32; do
    PFMSCAN_TILES_PER_BLOCK=$tpb python3 bench.py --workload c2 --mode hits --no-cpu-baseline "$@" 2>>gpurun_out/r3s/err.log | tail -1 > gpurun_out/r3s/tmp.json
    python3 - <<PY
import json
d=json.load(open("gpurun_out/r3s/tmp.json")); print("tpb $tpb  $*  %.4f ms" % d["ms_per_step"], d["config"].get("hits_per_step"))
PY
done
done
