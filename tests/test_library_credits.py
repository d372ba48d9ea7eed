"""The integer prefilter of the multi-PFM library kernel (pfmscan_library_api.hip, build_credits) must never drop a
hit.  No GPU needed: the credit table of one motif is built by the host code of libpfmscan and checked here
EXHAUSTIVELY -- every window of 4^m letter combinations -- against the reference's scoring rule
(_pwm.c:34-68: sequential fp64 sum, float32 cast; rnascan.py:263: strict `>`)."""
import itertools

import numpy as np
import pytest

from rnascan_amd import _lib


def all_windows(n_letters):
    return np.array(list(itertools.product(range(4), repeat=n_letters)), dtype=np.int64)


def exact_scores(T, codes):
    s = np.zeros(codes.shape[0], dtype=np.float64)
    for j in range(T.shape[0]):                       # sequential fp64 sum, as _pwm.c:36-64
        with np.errstate(invalid="ignore"):
            s = s + T[j, codes[:, j]]
    return s.astype(np.float32)


def credit_sums(credits, codes, m):
    npair = (m + 1) // 2
    tot = np.zeros(codes.shape[0], dtype=np.int64)
    for t in range(npair):
        idx = codes[:, 2 * t] | (codes[:, 2 * t + 1] << 2)
        tot += credits[t, idx].astype(np.int64)
    return tot


@pytest.mark.parametrize("bits", [16, 10])
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("inf_frac", [0.0, 0.2])
def test_prefilter_never_drops_a_hit(m, inf_frac, bits):
    """bits = 16: two credits per dword (k_letters_cred, wide library PFMs); bits = 10: three per dword, the library
    kernel's twelve-motifs-per-entry tables for widths up to 16"""
    rng = np.random.default_rng(31 * m + int(10 * inf_frac))
    top, flag = (1 << bits) - 1, 1 << (bits - 1)
    n_letters = m + (m & 1)                            # an odd width's last pair also sees the letter AFTER the window
    codes = all_windows(n_letters)
    for trial in range(6):
        T = np.full((m, 8), np.nan)
        T[:, :4] = rng.normal(0, 2.5, size=(m, 4)) * rng.choice([1.0, 1.0, 30.0])
        if inf_frac:
            T[:, :4][rng.random((m, 4)) < inf_frac] = -np.inf
        f = exact_scores(T, codes)
        fin = np.sort(f[np.isfinite(f)].astype(np.float64))
        thrs = [6.0, 0.0, -3.5, 1e4, -1e4, np.inf]
        if fin.size:
            thrs += [float(fin[int(q * (fin.size - 1))]) for q in (0.0, 0.5, 0.9, 0.99, 1.0)]            # ON scores
            thrs += [float(np.nextafter(np.float32(fin[int(0.9 * (fin.size - 1))]), np.float32(-np.inf)))]
        for thr in thrs:
            credits, slack = _lib.credit_table(T, thr, bits)
            tot = credit_sums(credits, codes, m)
            assert tot.max() <= top, "a sum would carry into the neighbouring motif's field"
            flagged = (tot & flag) != 0
            hit = f.astype(np.float64) > thr
            assert not (hit & ~flagged).any(), "prefilter dropped a hit (m=%d thr=%r)" % (m, thr)
            if np.isfinite(slack) and flagged.any():
                # and it is tight: every kept window scores within `slack` (+ the float32 rounding) of the threshold
                kept = f[flagged].astype(np.float64)
                assert (kept > thr - slack - 1e-5 * (1 + abs(thr)) - 2e-6 * np.abs(T[:, :4][np.isfinite(T[:, :4])]).sum()).all()


def test_prefilter_disabled_for_plus_inf_cells():
    T = np.full((4, 8), np.nan)
    T[:, :4] = np.random.default_rng(1).normal(0, 2, size=(4, 4))
    T[1, 2] = np.inf                                   # background 0 for a letter the PFM uses
    credits, slack = _lib.credit_table(T, 3.0)
    assert np.isinf(slack)
    codes = all_windows(4)
    assert ((credit_sums(credits, codes, 4) & 0x8000) != 0).all()          # everything goes to the exact pass


def test_wide_motifs_keep_sums_inside_their_fields():
    rng = np.random.default_rng(9)
    for m, bits in ((12, 16), (12, 10), (16, 10), (9, 0), (18, 0), (18, 16), (33, 16), (64, 16)):
        T = np.full((m, 8), np.nan)
        T[:, :4] = rng.normal(0, 3, size=(m, 4))
        used = bits or (10 if m <= 16 else 16)            # bits = 0: what k_library takes at this width
        for thr in (6.0, -50.0, 40.0):
            credits, slack = _lib.credit_table(T, thr, bits)
            assert int(credits.max(axis=1).astype(np.int64).sum()) <= (1 << used) - 1
            # random windows: hits are never dropped
            codes = rng.integers(0, 4, size=(200000, m + (m & 1)))
            f = exact_scores(T, codes)
            flagged = (credit_sums(credits, codes, m) & (1 << (used - 1))) != 0
            assert not ((f.astype(np.float64) > thr) & ~flagged).any()
            # best window of the motif: certainly kept when it is a hit
            best = np.argmax(T[:, :4], axis=1)[None, :]
            best = np.concatenate([best, np.zeros((1, m & 1), dtype=best.dtype)], axis=1)
            if float(exact_scores(T, best)[0]) > thr:
                assert (credit_sums(credits, best, m) & (1 << (used - 1))).all()


def quad_sums_of(credits, codes, m):
    """credit sum of every window under the FOUR-letter tables of k_letters_quad (rows of motif positions 4t .. 4t+3;
    the last row of a width that is no multiple of 4 also sees the letters AFTER the window)"""
    nq = (m + 3) // 4
    tot = np.zeros(codes.shape[0], dtype=np.int64)
    for t in range(nq):
        idx = codes[:, 4 * t] | (codes[:, 4 * t + 1] << 2) | (codes[:, 4 * t + 2] << 4) | (codes[:, 4 * t + 3] << 6)
        tot += credits[t, idx].astype(np.int64)
    return tot


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("inf_frac", [0.0, 0.2])
def test_quad_prefilter_never_drops_a_hit(m, inf_frac):
    """k_letters_quad's table (pfmscan_debug_quad_table): exhaustive over all 4^(4 ceil(m/4)) letter combinations"""
    rng = np.random.default_rng(77 * m + int(10 * inf_frac))
    n_letters = 4 * ((m + 3) // 4)
    codes = all_windows(n_letters)
    for trial in range(4):
        T = np.full((m, 8), np.nan)
        T[:, :4] = rng.normal(0, 2.5, size=(m, 4)) * rng.choice([1.0, 1.0, 30.0])
        if inf_frac:
            T[:, :4][rng.random((m, 4)) < inf_frac] = -np.inf
        f = exact_scores(T, codes)
        fin = np.sort(f[np.isfinite(f)].astype(np.float64))
        thrs = [6.0, 0.0, -3.5, 1e4, -1e4, np.inf]
        if fin.size:
            thrs += [float(fin[int(q * (fin.size - 1))]) for q in (0.0, 0.5, 0.9, 0.99, 1.0)]            # ON scores
            thrs += [float(np.nextafter(np.float32(fin[int(0.9 * (fin.size - 1))]), np.float32(-np.inf)))]
        for thr in thrs:
            credits, slack = _lib.quad_table(T, thr)
            tot = quad_sums_of(credits, codes, m)
            assert tot.max() <= 0xFFFF, "a sum would carry out of its 16-bit half"
            flagged = (tot & 0x8000) != 0
            hit = f.astype(np.float64) > thr
            assert not (hit & ~flagged).any(), "prefilter dropped a hit (m=%d thr=%r)" % (m, thr)
            if np.isfinite(slack) and flagged.any():
                kept = f[flagged].astype(np.float64)
                assert (kept > thr - slack - 1e-5 * (1 + abs(thr)) - 2e-6 * np.abs(T[:, :4][np.isfinite(T[:, :4])]).sum()).all()


def test_quad_tables_of_wide_motifs():
    rng = np.random.default_rng(19)
    for m in (9, 12, 13, 16, 18, 24, 31, 32):
        nq = (m + 3) // 4
        T = np.full((m, 8), np.nan)
        T[:, :4] = rng.normal(0, 3, size=(m, 4))
        for thr in (6.0, -50.0, 40.0):
            credits, slack = _lib.quad_table(T, thr)
            assert credits.shape == (nq, 256)
            assert int(credits.max(axis=1).astype(np.int64).sum()) <= 0xFFFF
            codes = rng.integers(0, 4, size=(200000, 4 * nq))
            f = exact_scores(T, codes)
            flagged = (quad_sums_of(credits, codes, m) & 0x8000) != 0
            assert not ((f.astype(np.float64) > thr) & ~flagged).any()
            # tighter than the two-letter table of the same motif: fewer rows, finer levels
            _, slack2 = _lib.credit_table(T, thr, 16)
            assert slack <= slack2 or not np.isfinite(slack2)
    with pytest.raises(ValueError):
        _lib.quad_table(np.full((33, 8), np.nan), 1.0)


# ---- single-letter credits of the generic-alphabet hits kernel (k_letters_cred8, pfmscan_letters8.hip) ----------------
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("special", ["none", "neg_inf", "nan", "pos_inf"])
def test_single_letter_credits_never_drop_a_hit(m, special):
    """every window over the 8 codes (7 letters + the foreign code) of width m, thresholds on and between scores: a window
    whose fp64 score exceeds the threshold (matrix.py:25-43 + the strict `>` of rnascan.py:263) always has bit 15 of its
    credit sum set; NaN / -inf cells and the foreign code never get credit; +inf cells switch the prefilter off"""
    rng = np.random.default_rng(77 * m + len(special))
    codes = np.array(list(itertools.product(range(8), repeat=m)), dtype=np.int64)
    for trial in range(5):
        T = np.full((m, 8), np.nan)
        T[:, :7] = rng.normal(-0.5, 2.5, size=(m, 7)) * rng.choice([1.0, 1.0, 25.0])
        r = rng.random((m, 7))
        if special == "neg_inf":
            T[:, :7][r < 0.2] = -np.inf
        elif special == "nan":
            T[:, :7][r < 0.15] = np.nan
        elif special == "pos_inf":
            T[:, :7][r < 0.1] = np.inf
        s = np.zeros(codes.shape[0])
        with np.errstate(invalid="ignore"):
            for j in range(m):
                s = s + T[j, codes[:, j]]                              # sequential fp64 sum
        fin = np.sort(s[np.isfinite(s)])
        thrs = [6.0, 0.0, -3.5, 1e4, -1e4, -1e300]
        if fin.size:
            thrs += [float(fin[int(q * (fin.size - 1))]) for q in (0.0, 0.5, 0.9, 0.99, 1.0)]
            thrs += [float(np.nextafter(fin[int(0.9 * (fin.size - 1))], -np.inf))]
        for thr in thrs:
            credits, mode = _lib.credit8_table(T, thr)
            if special == "pos_inf" and np.isinf(T[:, :7]).any() and (T[:, :7] == np.inf).any():
                assert mode == 3                                        # the exact kernel decides every window
                continue
            assert mode in (1, 2)
            tot = np.zeros(codes.shape[0], dtype=np.int64)
            for j in range(m):
                tot += credits[j, codes[:, j]].astype(np.int64)
            assert tot.max() <= 65535
            flagged = (tot & 0x8000) != 0
            with np.errstate(invalid="ignore"):
                hit = s > thr
            assert not (hit & ~flagged).any(), (m, special, thr)
            assert not flagged[(codes == 7).any(axis=1)].any()          # a foreign letter: NaN, never a hit, never kept
            if mode == 1 and hit.any() and np.isfinite(thr):            # the slack is small: kept windows lie near the threshold
                kept = s[flagged & np.isfinite(s)]
                span = float(np.nanmax(np.abs(T[:, :7][np.isfinite(T[:, :7])]))) * m + abs(thr) + 1.0
                assert kept.min() > thr - 0.02 * span - 1e-6


def test_single_letter_credits_dense_thresholds_are_flagged():
    """mode 2 <=> more than 1/32 of uniformly drawn windows would survive (the exact kernel runs instead)"""
    rng = np.random.default_rng(5)
    T = np.full((6, 8), np.nan)
    T[:, :7] = rng.normal(0, 1.0, size=(6, 7))
    assert _lib.credit8_table(T, -50.0)[1] == 2
    assert _lib.credit8_table(T, 4.0)[1] == 1
    assert _lib.credit8_table(T, 1e9)[1] == 1


# ---- single-letter credits of the structure-letter LIBRARY kernel (k_library8: rows padded to a multiple of 4) -----------------
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("special", ["none", "neg_inf", "nan", "pos_inf"])
def test_library8_credits_never_drop_a_hit(m, special):
    """every window over the 8 codes of width m -- and every code in the positions the PADDING rows look at (a lane's look-ups
    run over 4 ceil(m/4) positions): a window whose fp64 score exceeds the threshold always has bit 15 of its 16-bit credit sum
    set, whatever follows it; sums stay inside their 16-bit field; foreign letters inside the window never pass"""
    rng = np.random.default_rng(91 * m + len(special))
    rows = (m + 3) // 4 * 4
    codes = np.array(list(itertools.product(range(8), repeat=m)), dtype=np.int64)
    for trial in range(4):
        T = np.full((m, 8), np.nan)
        T[:, :7] = rng.normal(-0.5, 2.5, size=(m, 7)) * rng.choice([1.0, 1.0, 25.0])
        r = rng.random((m, 7))
        if special == "neg_inf":
            T[:, :7][r < 0.2] = -np.inf
        elif special == "nan":
            T[:, :7][r < 0.15] = np.nan
        elif special == "pos_inf":
            T[:, :7][r < 0.1] = np.inf
        s = np.zeros(codes.shape[0])
        with np.errstate(invalid="ignore"):
            for j in range(m):
                s = s + T[j, codes[:, j]]
        fin = np.sort(s[np.isfinite(s)])
        thrs = [6.0, 0.0, -3.5, 1e4, -1e4]
        if fin.size:
            thrs += [float(fin[int(q * (fin.size - 1))]) for q in (0.0, 0.5, 0.9, 0.99, 1.0)]
            thrs += [float(np.nextafter(fin[int(0.9 * (fin.size - 1))], -np.inf))]
        for thr in thrs:
            credits, slack = _lib.library8_credits(T, thr)
            assert credits.shape == (rows, 8)
            if (T[:, :7] == np.inf).any():
                assert np.isinf(slack) and (credits[0] == 0x8000).all()      # no prefilter: every window goes to the exact pass
                continue
            pad = credits[m:]
            assert (pad == pad[:, :1]).all()                               # a padding row gives every code the same credit ...
            tot = np.zeros(codes.shape[0], dtype=np.int64)
            for j in range(m):
                tot += credits[j, codes[:, j]].astype(np.int64)
            tot += int(pad[:, 0].astype(np.int64).sum())                   # ... so what follows the window cannot matter
            assert tot.max() <= 65535
            flagged = (tot & 0x8000) != 0
            with np.errstate(invalid="ignore"):
                hit = s > thr
            assert not (hit & ~flagged).any(), (m, special, thr)
            assert not flagged[(codes == 7).any(axis=1)].any()
