import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from rnascan_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(1)
n = 100000 * 3001
codes = rng.integers(0, 4, size=n).astype(np.uint8); codes[3000::3001] = 7
ctx.stage(codes)
for m in (12, 16, 18, 24, 32):
    T = np.full((m, 8), np.nan); P = rng.dirichlet(np.full(4, 0.5), size=m); T[:, :4] = np.log2((P + 0.01) / (1 + 0.04) / 0.25)
    mo = ctx.motif(letter_table=T)
    for thr in (6.0, 30.0):
        ctx.hits_staged(mo, thr, -np.inf)
        t = time.time()
        for _ in range(20): r = ctx.hits_staged(mo, thr, -np.inf)
        print("m=%d thr=%g: %.3f ms per call incl. read-back, %d hits" % (m, thr, (time.time() - t) / 20 * 1e3, len(r[0])))
    mo.close()
