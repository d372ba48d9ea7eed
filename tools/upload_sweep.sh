#!/bin/bash
# sweep of the uploader's knobs on the (c) warm-anonymous and (a) fresh-mapping cases of tools/mmap_upload_probe.py
for cfg in "0 64 16 1" "1 64 16 1" "1 64 16 0" "1 128 16 1" "1 32 16 1" "1 64 8 1" "1 256 16 1" "1 64 12 1"; do
  set -- $cfg
  echo "== UPLOAD=$1 PIECE_MB=$2 THREADS=$3 NT=$4"
  PFMSCAN_UPLOAD=$1 PFMSCAN_UPLOAD_PIECE_MB=$2 PFMSCAN_UPLOAD_THREADS=$3 PFMSCAN_UPLOAD_NT=$4 python3 tools/mmap_upload_probe.py 4 2>&1 | grep "(c) stage from anonymous memory  \|(a) stage from a fresh\|again"
done
