"""Property-based parity (hypothesis): random ragged record sets, widths, alphabets, thresholds and
PSSMs with -inf / +inf / NaN cells through the C ABI against the CPU oracle."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import assert_f32_bits_equal, assert_struct_close
from rnascan_amd import pack

pytestmark = pytest.mark.gpu

SPECIAL = [-np.inf, np.inf, np.nan, 0.0, -0.0, 1e-300, -1e300]


@st.composite
def cases(draw):
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    m = draw(st.integers(1, 40))
    n_rec = draw(st.integers(1, 12))
    lengths = [draw(st.integers(0, 300)) for _ in range(n_rec)]
    nletters = draw(st.sampled_from([4, 7]))
    foreign = draw(st.sampled_from([0.0, 0.02, 0.3]))
    special = draw(st.sampled_from([0.0, 0.05, 0.3]))
    dtype = draw(st.sampled_from([np.float32, np.float64]))
    thr = draw(st.sampled_from([-np.inf, -20.0, -3.0, 0.0, 2.5, 50.0]))
    codes, profs = [], []
    for L in lengths:
        c = rng.integers(0, nletters, size=L).astype(np.uint8)
        c[rng.random(L) < foreign] = 7
        p = rng.dirichlet(np.full(7, 0.3), size=L) if L else np.zeros((0, 7))
        p[p < 0.05] = 0.0
        codes.append(c)
        profs.append(p.astype(dtype))
    T = np.full((m, 8), np.nan)
    T[:, :nletters] = rng.normal(0, 3, size=(m, nletters))
    P = rng.normal(-1, 3, size=(m, 7))
    for A in (T[:, :nletters], P):
        mask = rng.random(A.shape) < special
        A[mask] = rng.choice(SPECIAL, size=int(mask.sum()))
    return pack.pack(codes, profs, profile_dtype=dtype), T, P, m, thr


# the default run replays the same 60 examples every time (derandomize); an exploration run sets
# PFMSCAN_HYPOTHESIS_EXAMPLES (and --hypothesis-seed) and draws fresh ones
@settings(max_examples=int(os.environ.get("PFMSCAN_HYPOTHESIS_EXAMPLES", "60")), deadline=None,
          derandomize="PFMSCAN_HYPOTHESIS_EXAMPLES" not in os.environ, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(case=cases())
def test_random_streams_match_the_oracle(ctx, oracle, case):
    s, T, P, m, thr = case
    want_seq = oracle.stream_seq(s.codes, T)
    want_st = oracle.stream_struct(s.profile, P)
    motif = ctx.motif(T, P)
    got_seq, got_st = ctx.scan_host(motif, s.codes, s.profile)
    assert_f32_bits_equal(got_seq, want_seq)
    assert_struct_close(got_st, want_st)
    # combined hits (fused or candidate-then-verify, whichever the library picks) == filtering the oracle scores
    pos, hs, ht = ctx.hits_host(motif, s.codes, s.profile, thr_seq=thr, thr_struct=thr)
    want = oracle.stream_hits(want_seq, want_st, thr, thr)
    assert np.array_equal(pos, want)
    assert_f32_bits_equal(hs, want_seq[want])
    assert_struct_close(ht, want_st[want])
    # letters only, fp64 out (matrix.py:25-43): exact
    lo = ctx.motif(letter_table=T)
    f64 = ctx.scan_letters_f64_host(lo, s.codes)
    ref = oracle.stream_letters_f64(s.codes, T)
    assert np.array_equal(np.isnan(f64), np.isnan(ref)) and np.array_equal(f64[~np.isnan(ref)], ref[~np.isnan(ref)])
    lo.close()
    motif.close()


@st.composite
def library_cases(draw):
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    m = draw(st.sampled_from([1, 2, 3, 5, 8, 11, 12, 16, 17, 24, 33, 64]))
    n = draw(st.integers(1, 40))
    n_rec = draw(st.integers(1, 10))
    lengths = [draw(st.integers(0, 400)) for _ in range(n_rec)]
    foreign = draw(st.sampled_from([0.0, 0.02, 0.3]))
    special = draw(st.sampled_from([0.0, 0.0, 0.05, 0.3]))
    struct = draw(st.booleans())
    dtype = draw(st.sampled_from([np.float32, np.float64]))
    codes, profs = [], []
    for L in lengths:
        c = rng.integers(0, 4, size=L).astype(np.uint8)
        c[rng.random(L) < foreign] = 7
        p = rng.dirichlet(np.full(7, 0.3), size=L) if L else np.zeros((0, 7))
        p[p < 0.05] = 0.0
        codes.append(c)
        profs.append(p.astype(dtype))
    T = np.full((n, m, 8), np.nan)
    T[:, :, :4] = rng.normal(0, 3, size=(n, m, 4)) * draw(st.sampled_from([1.0, 1.0, 100.0]))
    P = rng.normal(-1, 3, size=(n, m, 7)) if struct else None
    for A in ([T[:, :, :4]] + ([P] if struct else [])):
        mask = rng.random(A.shape) < special
        A[mask] = rng.choice([-np.inf, -np.inf, np.inf, 0.0, -0.0, 1e-300], size=int(mask.sum()))
    ts = rng.choice([-30.0, -5.0, 0.0, 3.0, 9.0, 40.0, np.inf], size=n)
    tt = rng.choice([-np.inf, -40.0, -10.0, 0.0], size=n)
    return pack.pack(codes, profs, profile_dtype=dtype), T, P, ts, tt


@settings(max_examples=int(os.environ.get("PFMSCAN_HYPOTHESIS_EXAMPLES", "60")), deadline=None,
          derandomize="PFMSCAN_HYPOTHESIS_EXAMPLES" not in os.environ, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
@given(case=library_cases())
def test_random_libraries_match_the_oracle(ctx, oracle, case):
    """the one-pass library kernel: random library sizes, widths of every template bucket, per-motif thresholds, -inf /
    +inf cells (motifs without prefilter), foreign letters, empty records, both profile storages"""
    s, T, P, ts, tt = case
    lib = ctx.library(T, P)
    pos, mo, sq, st = ctx.library_hits_host(lib, s.codes, s.profile if P is not None else None, ts, tt)
    lib.close()
    k0 = 0
    order = np.lexsort((mo, pos))
    assert np.array_equal(order, np.arange(len(pos)))             # sorted by (position, motif)
    for k in range(T.shape[0]):
        w_seq = oracle.stream_seq(s.codes, T[k])
        w_st = oracle.stream_struct(s.profile, P[k]) if P is not None else None
        want = oracle.stream_hits(w_seq, w_st, ts[k], tt[k] if P is not None else -np.inf)
        sel = mo == k
        assert np.array_equal(pos[sel], want), (k, ts[k], tt[k])
        assert_f32_bits_equal(sq[sel], w_seq[want])
        if P is not None:
            assert_struct_close(st[sel], w_st[want])
        k0 += len(want)
    assert k0 == len(pos)
