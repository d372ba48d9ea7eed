#!/bin/bash
# library-kernel builds side by side on one box: C5 (w = 12, both row types) and the wide buckets
# usage: tools/gpu_ab_lib_widths.sh lib1.so lib2.so ...  -> gpurun_out/r3_ab_lib/widths.jsonl
OUT=gpurun_out/r3_ab_lib
mkdir -p $OUT
: > $OUT/widths.jsonl
probe() { lib=$1; shift; PFMSCAN_LIB=$(pwd)/rnascan_amd/$lib python3 tools/c5_probe.py "$@" 2>>$OUT/err.log | tail -1 >> $OUT/widths.jsonl; }
THR="--thr-struct -10.5"
for round in 1 2; do
  for lib in "$@"; do
    probe $lib --width 12 $THR
    probe $lib --width 12 $THR --profile-dtype float64
    probe $lib --width 18 --motifs 128 $THR
    probe $lib --width 24 --motifs 128 $THR --profile-dtype float64
    probe $lib --width 40 --motifs 64 $THR
  done
done
python3 - <<'PY'
import json
for ln in open("gpurun_out/r3_ab_lib/widths.jsonl"):
    d = json.loads(ln)
    print("%-22s w=%2d %-8s motifs=%3d ms=%8.3f hits=%d" % (d["lib"], d["width"], d["profile_dtype"], d["motifs"], d["ms"], d["hits"]))
PY
