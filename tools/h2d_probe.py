#!/usr/bin/env python3
"""time pfmscan_stage (host -> device) for a large packed profile"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from rnascan_amd import _lib
ctx = _lib.Context(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
prof = np.random.default_rng(0).random((n, 7), dtype=np.float32)
codes = np.zeros(n, dtype=np.uint8)
for rep in range(3):
    t = time.perf_counter()
    ctx.stage(codes, prof)
    dt = time.perf_counter() - t
    print("stage %.2f GB in %.3f s = %.1f GB/s" % ((prof.nbytes + codes.nbytes) / 1e9, dt, (prof.nbytes + codes.nbytes) / 1e9 / dt))
