#!/bin/bash
# A/B of library-kernel builds on one box, one call: r2 build vs current vs the 1024-thread variant of the wide bucket.
# usage: tools/gpu_ab_r3_lib.sh  -> gpurun_out/r3_ab_lib/ab.jsonl
OUT=gpurun_out/r3_ab_lib
mkdir -p $OUT
: > $OUT/ab.jsonl
probe() { lib=$1; shift; PFMSCAN_LIB=$(pwd)/rnascan_amd/$lib python3 tools/c5_probe.py "$@" 2>>$OUT/err.log | tail -1 >> $OUT/ab.jsonl; }
THR="--thr-struct -10.5"
for round in 1 2; do
  for lib in libpfmscan_r2.so libpfmscan.so; do
    probe $lib --width 12 $THR
    probe $lib --width 12 $THR --profile-dtype float64
  done
  for lib in libpfmscan_r2.so libpfmscan_b1024.so libpfmscan.so; do
    probe $lib --width 18 --motifs 128 $THR
    probe $lib --width 24 --motifs 128 $THR
    probe $lib --width 24 --motifs 128 $THR --profile-dtype float64
    probe $lib --width 32 --motifs 128 $THR
  done
done
python3 - <<'PY'
import json
for ln in open("gpurun_out/r3_ab_lib/ab.jsonl"):
    d = json.loads(ln)
    print("%-22s w=%2d %-8s motifs=%3d ms=%8.3f hits=%d" % (d["lib"], d["width"], d["profile_dtype"], d["motifs"], d["ms"], d["hits"]))
PY
