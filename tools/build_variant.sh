#!/bin/bash
# a variant of libpfmscan.so with one source recompiled under extra flags (A/B and diagnostic builds):
#   tools/build_variant.sh <tag> <source.hip> <flags...>   ->  rnascan_amd/libpfmscan_<tag>.so   (select with PFMSCAN_LIB)
set -e
tag=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
python3 -c "import sys; sys.path.insert(0, '$root'); from rnascan_amd import build; build.build_lib()" > /dev/null
obj=/tmp/variant_${tag}_$(basename "$src" .hip).o
hipcc ${VARIANT_OPT:--O3} --offload-arch=gfx950 -std=c++17 -fPIC -fno-fast-math -Wall "$@" -c "$root/rnascan_amd/csrc/$src" -o "$obj"
objs=""
for o in "$root"/rnascan_amd/csrc/_obj/*.o; do
  if [ "$(basename "$o" .o)" = "$(basename "$src" .hip)" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -fPIC -shared -o "$root/rnascan_amd/libpfmscan_${tag}.so" $objs
echo "$root/rnascan_amd/libpfmscan_${tag}.so"
