#!/bin/bash
# library kernel at C5 and the wide buckets, with and without an environment switch: tools/gpu_ab_lib_env.sh "ENV=1"
OUT=gpurun_out/r3_ab_lib
mkdir -p $OUT
: > $OUT/env.jsonl
probe() { python3 tools/c5_probe.py "$@" 2>>$OUT/err.log | tail -1 >> $OUT/env.jsonl; }
THR="--thr-struct -10.5"
for round in 1 2; do
  for e in "" "$1"; do
    if [ -n "$e" ]; then export $e; else unset ${1%%=*}; fi
    echo "{\"env\": \"$e\"}" >> $OUT/env.jsonl
    probe --width 12 $THR
    probe --width 12 $THR --profile-dtype float64
    probe --width 18 --motifs 128 $THR
    probe --width 24 --motifs 128 $THR --profile-dtype float64
    probe --width 40 --motifs 64 $THR
    probe --width 12 --motifs 1000 $THR
  done
done
python3 - <<'PY'
import json
env = ""
for ln in open("gpurun_out/r3_ab_lib/env.jsonl"):
    d = json.loads(ln)
    if "env" in d: env = d["env"] or "(default)"; continue
    print("%-28s w=%2d %-8s motifs=%4d ms=%8.3f hits=%d" % (env, d["width"], d["profile_dtype"], d["motifs"], d["ms"], d["hits"]))
PY
