#!/usr/bin/env python3
"""time a seq-only PFM library over one staged stream through the host API (what the CLI does per motif)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from rnascan_amd import _lib

def main():
    n_rec, L, w, n_motifs, thr = 100000, 3000, 8, 16, 6.0
    rng = np.random.default_rng(1)
    codes = rng.integers(0, 4, size=n_rec * (L + 1), dtype=np.uint8)
    codes[L::L + 1] = 7
    ctx = _lib.Context(0)
    t = time.perf_counter(); ctx.stage(codes, None); print("stage %.1f ms" % ((time.perf_counter() - t) * 1e3))
    motifs = [ctx.motif(bench.make_pssms(w, "finite", seed=1000 + k)[0], None) for k in range(n_motifs)]
    for thr_k in (thr, 9.0, 30.0):
        for rep in range(2):
            t = time.perf_counter()
            tot = 0
            for mo in motifs:
                pos, sq, _ = ctx.hits_staged(mo, thr_seq=thr_k)
                tot += pos.size
            dt = time.perf_counter() - t
            print("thr %.0f rep %d: %d motifs, %d hits, %.2f ms per motif" % (thr_k, rep, n_motifs, tot, dt / n_motifs * 1e3))
    if hasattr(ctx, "hits_library_staged"):
        for rep in range(2):
            t = time.perf_counter()
            res = ctx.hits_library_staged(motifs, thr_seq=thr)
            dt = time.perf_counter() - t
            print("library rep %d: %d hits, %.2f ms per motif" % (rep, sum(r[0].size for r in res), dt / n_motifs * 1e3))
        for (p1, s1), mo in zip(res, motifs):
            p0, s0, _ = ctx.hits_staged(mo, thr_seq=thr)
            assert np.array_equal(p0, p1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32))
        print("library == per-motif: ok")

main()
