"""Thresholded STRUCTURE hits are the oracle's hit set exactly, also when the threshold sits ON a score or one ulp beside it.

The kernels' fast structure score (a multiply and six FMAs per row, chained into the window sum when the PSSM is finite)
differs from the reference-order score (rnascan.py:306: every product and every addition rounded, k ascending -- what
oracle/pfm_oracle.c restates and the goldens pin) in the last bits.  Scores only have to agree to 1e-6, hit POSITIONS are
bit-exact by contract: every hits path re-scores a window whose fast score lies within a rigorous band of the threshold in
the rounded order and compares (and reports) that value (rnascan_amd/csrc/pfmscan_exact.hpp).  The mirror of
test_prefilter_thresholds_on_existing_scores for the structure side, on every path that compares a structure score:
k_profile / k_profile_fixed (fused), k_struct_at (two-phase), k_wide, k_library phase B, k_profile_lib."""
import numpy as np
import pytest

from test_gpu_parity import rand_stream, rand_struct_pssm, rand_table

pytestmark = pytest.mark.gpu


def _picks(want_st, got_st, eligible, n, rng, need_differ=1):
    """windows whose FAST score differs from the oracle's in the last bits (so a compare of the fast score could go
    wrong) among the eligible ones, in each direction when there are; filled up with other eligible windows.  With float32
    rows and -inf cells (most row-dots then have few terms) the two orders agree almost everywhere: `need_differ` is what
    the case must offer at least for the test to prove something."""
    fin = np.isfinite(want_st) & eligible & (np.abs(want_st) < 1e300)
    differ = fin & (got_st != want_st)
    assert differ.sum() >= need_differ, "the fast path equals the oracle everywhere: this test would prove nothing"
    above = np.flatnonzero(differ & (got_st > want_st))
    below = np.flatnonzero(differ & (got_st < want_st))
    out = []
    for pool in (above, below):
        if pool.size:
            out.extend(rng.choice(pool, size=min(pool.size, (n + 1) // 2), replace=False).tolist())
    rest = np.setdiff1d(np.flatnonzero(fin), np.array(out, dtype=np.int64))
    if len(out) < n and rest.size:
        out.extend(rng.choice(rest, size=min(rest.size, n - len(out)), replace=False).tolist())
    if not out:                                      # -inf cells everywhere: any window with a finite score will do
        out = rng.choice(np.flatnonzero(np.isfinite(want_st)), size=n, replace=False).tolist()
    return out[:n]


def _thresholds(s):
    return (float(s), float(np.nextafter(s, -np.inf)), float(np.nextafter(s, np.inf)))


def _check_single(ctx, oracle, motif, s, want_seq, want_st, thr_seq, picks, with_codes):
    for p in picks:
        for thr in _thresholds(want_st[p]):
            pos, sq, st = ctx.hits_host(motif, s.codes if with_codes else None, s.profile, thr_seq=thr_seq, thr_struct=thr)
            want_pos = oracle.stream_hits(want_seq if with_codes else None, want_st, thr_seq, thr)
            assert np.array_equal(pos, want_pos), (p, thr, pos.size, want_pos.size)
            at = np.flatnonzero(pos == p)
            if thr < want_st[p]:                     # one ulp below: p is a hit, and its reported score is the re-scored one
                assert at.size == 1 and st[at[0]] == want_st[p]
            else:                                    # ON the score (strict >) or above it: p is out
                assert at.size == 0


@pytest.mark.parametrize("m", [7, 12, 18, 24])                   # generic (< 9, > 18) and fixed-width k_profile
@pytest.mark.parametrize("inf_frac", [0.0, 0.15])                # the FINITE (chained) and the per-row nan_to_num form
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_structure_only_thresholds_on_and_beside_scores(ctx, oracle, m, inf_frac, dtype):
    rng = np.random.default_rng(500 + m + (7 if inf_frac else 0))
    s = rand_stream(rng, 12, 200, 2500, dtype=dtype)
    P = rand_struct_pssm(rng, m, min(inf_frac, 1.0 / m))      # (wide PFMs: a few -inf cells, or no window has a finite score)
    motif = ctx.motif(None, P)
    want_st = oracle.stream_struct(s.profile, P)
    _, got_st = ctx.scan_host(motif, None, s.profile)
    picks = _picks(want_st, got_st, np.ones(want_st.size, bool), 4, rng, need_differ=0 if inf_frac else 4)
    _check_single(ctx, oracle, motif, s, None, want_st, -np.inf, picks, with_codes=False)
    motif.close()


@pytest.mark.parametrize("two_phase", [1, 0])                    # k_struct_at at the letters pass's hits / the fused k_profile pass
@pytest.mark.parametrize("m", [8, 12])
def test_combined_thresholds_on_and_beside_scores(oracle, monkeypatch, two_phase, m):
    from rnascan_amd import _lib
    monkeypatch.setenv("PFMSCAN_TWO_PHASE", str(two_phase))
    rng = np.random.default_rng(900 + m)
    s = rand_stream(rng, 14, 300, 2500)
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m)
    with _lib.Context(0) as c:
        motif = c.motif(T, P)
        want_seq, want_st = oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P)
        _, got_st = c.scan_host(motif, s.codes, s.profile)
        thr_seq = float(np.quantile(want_seq[np.isfinite(want_seq)], 0.985))   # selective: the pilot keeps the two-phase route
        picks = _picks(want_st, got_st, want_seq > thr_seq, 4, rng)
        _check_single(c, oracle, motif, s, want_seq, want_st, thr_seq, picks, with_codes=True)
        motif.close()


def test_wide_kernel_thresholds_on_and_beside_scores(ctx, oracle):
    """PFMs wider than k_profile takes (> 180 rows) run the plain one-thread-per-window kernel"""
    rng = np.random.default_rng(77)
    m = 200
    s = rand_stream(rng, 5, 600, 1500)
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m)
    motif = ctx.motif(T, P)
    want_seq, want_st = oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P)
    _, got_st = ctx.scan_host(motif, s.codes, s.profile)
    picks = _picks(want_st, got_st, np.isfinite(want_seq), 3, rng)
    _check_single(ctx, oracle, motif, s, want_seq, want_st, -np.inf, picks, with_codes=True)
    motif.close()


@pytest.mark.parametrize("inf_frac", [0.0, 0.1])
def test_library_thresholds_on_and_beside_scores(ctx, oracle, inf_frac):
    """k_library phase B: per-motif thresholds planted on / beside the oracle's score of one window of that motif"""
    rng = np.random.default_rng(41 + (1 if inf_frac else 0))
    n, m = 14, 12
    s = rand_stream(rng, 12, 300, 2500)
    LT = np.stack([rand_table(rng, m) for _ in range(n)])
    LP = np.stack([rand_struct_pssm(rng, m, inf_frac) for _ in range(n)])
    lib = ctx.library(LT, LP)
    want_seq = [oracle.stream_seq(s.codes, LT[k]) for k in range(n)]
    want_st = [oracle.stream_struct(s.profile, LP[k]) for k in range(n)]
    thr_seq = np.array([np.quantile(w[np.isfinite(w)], 0.9) for w in want_seq])
    planted = []
    for k in range(n):
        mo = ctx.motif(LT[k], LP[k])
        _, got_st = ctx.scan_host(mo, s.codes, s.profile)
        mo.close()
        planted.append(_picks(want_st[k], got_st, want_seq[k] > thr_seq[k], 1, rng, need_differ=0 if inf_frac else 1)[0])
    for which in range(3):                                        # ON the score, one ulp below, one ulp above -- every motif at once
        thr_st = np.array([_thresholds(want_st[k][planted[k]])[which] for k in range(n)])
        pos, mot, sq, st = ctx.library_hits_host(lib, s.codes, s.profile, thr_seq, thr_st)
        for k in range(n):
            sel = mot == k
            want_pos = oracle.stream_hits(want_seq[k], want_st[k], thr_seq[k], thr_st[k])
            assert np.array_equal(pos[sel], want_pos), (which, k, int(sel.sum()), want_pos.size)
            at = np.flatnonzero(sel & (pos == planted[k]))
            if which == 1 and want_seq[k][planted[k]] > thr_seq[k]:
                assert at.size == 1 and st[at[0]] == want_st[k][planted[k]]
            else:
                assert at.size == 0
    lib.close()


@pytest.mark.parametrize("inf_frac", [0.0, 0.1])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_structure_library_thresholds_on_and_beside_scores(ctx, oracle, inf_frac, dtype):
    """k_profile_lib (structure-only libraries)"""
    rng = np.random.default_rng(61 + (1 if inf_frac else 0))
    n, m = 9, 12
    s = rand_stream(rng, 8, 300, 2500, dtype=dtype)
    LP = np.stack([rand_struct_pssm(rng, m, inf_frac) for _ in range(n)])
    lib = ctx.library(None, LP)
    want_st = [oracle.stream_struct(s.profile, LP[k]) for k in range(n)]
    planted = []
    for k in range(n):
        mo = ctx.motif(None, LP[k])
        _, got_st = ctx.scan_host(mo, None, s.profile)
        mo.close()
        top = want_st[k] > np.quantile(want_st[k][np.isfinite(want_st[k])], 0.9)
        planted.append(_picks(want_st[k], got_st, top, 1, rng, need_differ=0 if inf_frac else 1)[0])
    for which in range(3):
        thr_st = np.array([_thresholds(want_st[k][planted[k]])[which] for k in range(n)])
        pos, mot, _, st = ctx.library_hits_host(lib, None, s.profile, None, thr_st)
        for k in range(n):
            sel = mot == k
            want_pos = oracle.stream_hits(None, want_st[k], -np.inf, thr_st[k])
            assert np.array_equal(pos[sel], want_pos), (which, k, int(sel.sum()), want_pos.size)
            at = np.flatnonzero(sel & (pos == planted[k]))
            if which == 1:
                assert at.size == 1 and st[at[0]] == want_st[k][planted[k]]
            else:
                assert at.size == 0
    lib.close()
