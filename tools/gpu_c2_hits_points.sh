#!/bin/bash
# the single-motif hits kernels at the points DESIGN quotes (C2 widths / thresholds, C3 two-phase); optional: ENV=.. pairs first
for kv in $1; do export $kv; done
for a in "--width 8 --minscore-seq 6" "--width 8 --minscore-seq 30" "--width 4 --minscore-seq 2" "--width 12 --minscore-seq 6" "--width 18 --minscore-seq 8" "--width 32 --minscore-seq 10"; do
  python3 bench.py --workload c2 --mode hits --no-cpu-baseline $a 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $a', round(d['ms_per_step'],4), d['config'].get('hits_per_step'))"
done
python3 bench.py --mode hits2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 hits2', round(d['ms_per_step'],4), d['config'].get('hits_per_step'))"
