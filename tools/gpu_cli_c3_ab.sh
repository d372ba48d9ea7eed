#!/bin/bash
# C3's and C5's command lines (tools/cli_e2e.py, big inputs only) with the staged upload reading the FILE behind the store's
# mapping (default) and the mapping itself (PFMSCAN_UPLOAD_NO_PREAD=1), same box, two rounds
export TMPDIR=/dev/shm
mkdir -p gpurun_out/r3e
for round in 1 2; do
  for e in "" "PFMSCAN_UPLOAD_NO_PREAD=1"; do
    echo "== ${e:-pread (default)}"
    env $e CLI_E2E_LIBRARY=256 python3 tools/cli_e2e.py 2000 3000 100000 store 2> gpurun_out/r3e/ab.err | grep "^big\|^library"
    rm -rf /dev/shm/tmp*
  done
done
