#!/usr/bin/env python3
"""host-buffer entry points end to end: pfmscan_stage (H2D) and pfmscan_scan_host / hits_host
(H2D + kernel + D2H) on a C3-shaped batch of `records` records x 3 kb"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench
from rnascan_amd import _lib

records = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L, m = 3000, 12
ctx = _lib.Context(0)
rng = np.random.default_rng(0)
n = records * (L + 1)
prof = rng.random((n, 7), dtype=np.float32)
prof /= prof.sum(axis=1, keepdims=True)
codes = rng.integers(0, 4, size=n).astype(np.uint8)
codes[L::L + 1] = 7
table, spssm = bench.make_pssms(m)
motif = ctx.motif(table, spssm)
windows = records * (L - m + 1)
for rep in range(3):
    t = time.perf_counter()
    ctx.stage(codes, prof)
    dt = time.perf_counter() - t
    print("stage      %.2f GB in %.3f s = %.1f GB/s" % ((prof.nbytes + codes.nbytes) / 1e9, dt, (prof.nbytes + codes.nbytes) / 1e9 / dt))
for rep in range(3):
    t = time.perf_counter()
    sq, st = ctx.scan_host(motif, codes, prof)
    dt = time.perf_counter() - t
    print("scan_host  all-scores end to end %.3f s = %.3g windows/s" % (dt, windows / dt))
for rep in range(3):
    t = time.perf_counter()
    pos, a, b = ctx.hits_host(motif, codes, prof, 6.0, 6.0)
    dt = time.perf_counter() - t
    print("hits_host  -m 6 end to end %.3f s = %.3g windows/s (%d hits)" % (dt, windows / dt, len(pos)))
