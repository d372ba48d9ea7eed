#!/bin/bash
# the same bench.py line with several builds of the library, two rounds: tools/gpu_ab_libs.sh "lib1.so lib2.so" <bench args>
LIBS=$1; shift
for round in 1 2; do
for lib in $LIBS; do
  PFMSCAN_LIB=$(pwd)/rnascan_amd/$lib python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib  $*  ms_per_step %.4f  frac %s  parity %s' % (d['ms_per_step'], (d.get('roofline') or {}).get('frac'), d.get('parity_on_sample')))"
done
done
