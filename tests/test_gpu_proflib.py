"""Structure-only PFM library in ONE pass over the profile (k_profile_lib, SURVEY 8f N1 for `-q library avgdir/`)
through the C ABI, against the CPU oracle run one motif at a time (rnascan.py:302-310 per motif): the hit set
{p : struct_k(p) > thr[k]} of EVERY motif is identical and the scores are within 1e-6 -- and bit-identical to the
single-motif kernel's (k_profile), whose operation order the library kernel keeps."""
import numpy as np
import pytest

from conftest import assert_struct_close
from test_gpu_library import _below_max, _clear_of
from test_gpu_parity import rand_stream, rand_struct_pssm

pytestmark = pytest.mark.gpu


def oracle_hits(oracle, profile, P, thr):
    pos, mo, st_l = [], [], []
    for k in range(P.shape[0]):
        st = oracle.stream_struct(profile, P[k])
        p = oracle.stream_hits(None, st, -np.inf, thr[k])
        pos.append(p)
        mo.append(np.full(p.size, k, dtype=np.int32))
        st_l.append(st[p])
    pos, mo, st_l = np.concatenate(pos), np.concatenate(mo), np.concatenate(st_l)
    order = np.lexsort((mo, pos))
    return pos[order], mo[order], st_l[order]


def thresholds(oracle, profile, P, q):
    thr = np.empty(P.shape[0])
    for k in range(P.shape[0]):
        st = oracle.stream_struct(profile, P[k])
        fin = st[np.isfinite(st) & (np.abs(st) < 1e9)]
        thr[k] = _clear_of(fin, _below_max(fin, q))
    return thr


def make_pssms(rng, n, m, inf_every=0):
    """n structure PSSMs; with inf_every > 0 every inf_every-th motif has -inf cells (finite and generic motifs mixed in
    one library: the kernel picks the form per motif)"""
    return np.stack([rand_struct_pssm(rng, m, inf_frac=0.15 if (inf_every and k % inf_every == 0) else 0.0) for k in range(n)])


@pytest.mark.parametrize("n,m,dtype,inf_every", [(1, 12, np.float32, 0), (2, 1, np.float32, 0), (16, 12, np.float32, 3),
                                                 (16, 12, np.float64, 3), (7, 5, np.float64, 1), (33, 18, np.float32, 2),
                                                 (5, 64, np.float32, 2), (300, 7, np.float32, 4), (12, 13, np.float64, 0)])
def test_struct_library_hits_match_oracle_per_motif(ctx, oracle, n, m, dtype, inf_every):
    rng = np.random.default_rng(1000 * n + m)
    s = rand_stream(rng, 24, 0, 900, dtype=dtype)
    P = make_pssms(rng, n, m, inf_every)
    thr = thresholds(oracle, s.profile, P, 0.9)
    lib = ctx.library(None, P)
    pos, mo, sq, st = ctx.library_hits_host(lib, None, s.profile, None, thr)
    wp, wm, wst = oracle_hits(oracle, s.profile, P, thr)
    assert len(wp) > 20 and sq is None
    assert np.array_equal(pos, wp) and np.array_equal(mo, wm)
    assert_struct_close(st, wst, tol=1e-6)
    info = lib.info()
    assert info["n_motifs"] == n and info["m"] == m and info["passes"] == 1 and info["motifs_per_pass"] == n
    # the same numbers the single-motif kernel gives, bit for bit
    ctx.stage(None, s.profile)
    for k in sorted(set([0, n // 2, n - 1])):
        motif = ctx.motif(None, P[k])
        _, full = ctx.scan_staged(motif)
        motif.close()
        sel = mo == k
        assert np.array_equal(st[sel].view(np.uint64), full[pos[sel]].view(np.uint64))
    lib.close()


def test_struct_library_tile_edges_and_stream_end(ctx, oracle):
    """stream lengths around the 1280-position tile: the windows of the last tile that run over the end never hit"""
    rng = np.random.default_rng(8)
    P = make_pssms(rng, 6, 12, 0)
    lib = ctx.library(None, P)
    for n_pos in (1, 11, 12, 13, 1279, 1280, 1281, 1291, 1292, 2560, 2571, 3000):
        prof = rng.dirichlet(np.full(7, 0.3), size=n_pos).astype(np.float32)
        thr = np.full(6, -1e300)                           # every window that has a score is a hit
        pos, mo, _, st = ctx.library_hits_host(lib, None, prof, None, thr)
        wp, wm, wst = oracle_hits(oracle, prof, P, thr)
        assert len(wp) == 6 * max(0, n_pos - 12 + 1)
        assert np.array_equal(pos, wp) and np.array_equal(mo, wm)
        assert_struct_close(st, wst, tol=1e-6)
    lib.close()


def test_struct_library_dense_hits_flush_the_wave_queues(ctx, oracle):
    """a threshold below every score: each wave pushes 64 hits per window slot and motif, far more than its queue holds"""
    rng = np.random.default_rng(3)
    s = rand_stream(rng, 9, 500, 900)
    P = make_pssms(rng, 5, 9, 0)
    lib = ctx.library(None, P)
    thr = np.full(5, -1e6)
    got = ctx.library_hits_host(lib, None, s.profile, None, thr)
    wp, wm, wst = oracle_hits(oracle, s.profile, P, thr)
    assert len(wp) == 5 * (s.n_pos - 9 + 1)
    assert np.array_equal(got[0], wp) and np.array_equal(got[1], wm)
    assert_struct_close(got[3], wst, tol=1e-6)
    # capacity protocol: too small a buffer reports the need and writes nothing
    from rnascan_amd import _lib
    with pytest.raises(_lib.CapacityError) as e:
        ctx.library_hits_staged(lib, None, thr, capacity=100)
    assert e.value.required >= len(wp)
    lib.close()


def test_struct_library_with_nonfinite_profile_values(ctx, oracle):
    """NaN / inf in the PROFILE itself: finite motifs leave their fast path for the exact per-row sum (nan_to_num per row)"""
    rng = np.random.default_rng(4)
    s = rand_stream(rng, 12, 200, 600, dtype=np.float64)
    prof = s.profile.copy()
    idx = rng.integers(0, prof.shape[0], size=40)
    prof[idx[:15], rng.integers(0, 7, size=15)] = np.nan
    prof[idx[15:30], rng.integers(0, 7, size=15)] = np.inf
    prof[idx[30:], rng.integers(0, 7, size=10)] = -np.inf
    P = make_pssms(rng, 8, 10, 3)
    thr = np.full(8, -20.0)
    lib = ctx.library(None, P)
    pos, mo, _, st = ctx.library_hits_host(lib, None, prof, None, thr)
    wp, wm, wst = oracle_hits(oracle, prof, P, thr)
    assert len(wp) > 100
    assert np.array_equal(pos, wp) and np.array_equal(mo, wm)
    assert_struct_close(st, wst, tol=1e-6)
    lib.close()


def test_struct_library_argument_errors(ctx):
    from rnascan_amd import _lib
    P = np.zeros((3, 4, 7))
    lib = ctx.library(None, P)
    with pytest.raises(ValueError):
        ctx.library_hits_host(lib, None, np.zeros((10, 7), dtype=np.float32), None, np.array([0.0, np.nan, 0.0]))
    with pytest.raises(ValueError):
        ctx.library(None, None)
    with pytest.raises(ValueError):
        ctx.library(None, np.zeros((2, 65, 7)))
    lib.close()


def test_scanner_struct_library_is_one_pass_and_equals_per_motif_tables(ctx, oracle, tmp_path):
    """scanner layer: a structure PFM library over averaged-structure profiles = the per-motif tables, merged"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from collections import OrderedDict
    from engines import OracleEngine
    from rnascan_amd import pack, pssm, scanner
    rng = np.random.default_rng(12)
    named = []
    for i in range(14):
        L = int(rng.integers(5, 400))
        p = rng.dirichlet(np.full(7, 0.3), size=L)
        named.append(("rec%d" % i, list("BEHLMRT"), p))
    lib = OrderedDict()
    for k in range(9):
        m = 12 if k % 3 else 7
        counts = rng.dirichlet(np.full(7, 0.5), size=m)
        d = OrderedDict((l, counts[:, c]) for c, l in enumerate(pack.STRUCT_COLUMNS))
        lib["motif%02d" % k] = pssm.PSSM(pack.STRUCT_COLUMNS, pssm.log_odds(pssm.normalize(d, 0.01), None))
    eng = scanner.HipEngine(0)
    calls = []
    real = eng.library_hits
    eng.library_hits = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    got = scanner.scan_profiles(eng, named, lib, -8.0, "aligned", np.float64)
    want = scanner.scan_profiles(OracleEngine(), named, lib, -8.0, "aligned", np.float64)
    eng.close()
    assert len(calls) == 2                                   # one pass per PFM width
    assert len(want) > 50
    assert list(got.columns) == list(want.columns)
    for c in ("Sequence_ID", "Motif_ID", "Start", "End"):
        assert list(got[c]) == list(want[c])
    assert np.abs(got["LogOdds"].to_numpy() - want["LogOdds"].to_numpy()).max() <= 1e-6
