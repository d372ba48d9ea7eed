"""Multi-PFM library scan (k_library, SURVEY 8f N1 / BASELINE config 5) through the C ABI, against the CPU
oracle run one motif at a time: for EVERY motif of the library the hit set {p : seq_k(p) > thr_seq[k] and
struct_k(p) > thr_struct[k]} is identical, float32 sequence scores are bit-exact, structure scores within 1e-6.
The kernel's fp16 prefilter may only drop windows that cannot be hits, so the thresholds here are put where it
hurts: at quantiles of the score distribution and exactly ON the scores of existing windows (strict `>`)."""
import numpy as np
import pytest

from conftest import assert_f32_bits_equal, assert_struct_close
from test_gpu_parity import rand_stream, rand_struct_pssm, rand_table

pytestmark = pytest.mark.gpu


def make_library(rng, n, m, inf_frac=0.0, struct=True):
    T = np.stack([rand_table(rng, m, 4, inf_frac=inf_frac / 3) for _ in range(n)])
    P = np.stack([rand_struct_pssm(rng, m, inf_frac=inf_frac) for _ in range(n)]) if struct else None
    return T, P


def oracle_library_hits(oracle, s, T, P, thr_seq, thr_struct):
    """(pos, motif, seq, struct) arrays sorted by (pos, motif) from per-motif oracle scans"""
    n = T.shape[0]
    pos, mo, sq_l, st_l = [], [], [], []
    for k in range(n):
        sq = oracle.stream_seq(s.codes, T[k])
        st = oracle.stream_struct(s.profile, P[k]) if P is not None else None
        p = oracle.stream_hits(sq, st, thr_seq[k], thr_struct[k] if P is not None else -np.inf)
        pos.append(p)
        mo.append(np.full(p.size, k, dtype=np.int32))
        sq_l.append(sq[p])
        st_l.append(st[p] if st is not None else np.full(p.size, np.nan))
    pos, mo, sq_l, st_l = np.concatenate(pos), np.concatenate(mo), np.concatenate(sq_l), np.concatenate(st_l)
    order = np.lexsort((mo, pos))
    return pos[order], mo[order], sq_l[order], st_l[order]


def check(got, want, has_struct):
    pos, mo, sq, st = got
    wp, wm, wsq, wst = want
    assert len(pos) == len(wp), "hit count %d, oracle %d" % (len(pos), len(wp))
    if not len(wp):
        return
    assert np.array_equal(pos, wp) and np.array_equal(mo, wm)
    assert_f32_bits_equal(sq, wsq)
    if has_struct:
        assert_struct_close(st, wst, tol=1e-6)


def quantile_thresholds(oracle, s, T, P, q_seq, q_struct):
    """per-motif thresholds at quantiles of the finite scores (so every motif has a realistic number of hits)"""
    n = T.shape[0]
    ts, tt = np.empty(n), np.full(n, -np.inf)
    for k in range(n):
        sq = oracle.stream_seq(s.codes, T[k]).astype(np.float64)
        ts[k] = _below_max(sq[np.isfinite(sq)], q_seq)
        if P is not None:
            st = oracle.stream_struct(s.profile, P[k])
            tt[k] = _clear_of(st[np.isfinite(st) & (np.abs(st) < 1e9)], _below_max(st[np.isfinite(st) & (np.abs(st) < 1e9)], q_struct))
    return ts, tt


def _clear_of(fin, t):
    """structure scores are compared to 1e-6, so a structure threshold must not sit ON a score (the kernel's FMA
    chain and the oracle's mul + add differ in the last bit): move it to the middle of the gap between scores"""
    if not fin.size:
        return t
    v = np.unique(fin)
    i = int(np.searchsorted(v, t, side="right"))         # v[i-1] <= t < v[i]
    lo = v[i - 1] if i > 0 else t - 1.0
    hi = v[i] if i < v.size else t + 1.0
    return float(lo + (hi - lo) / 2)


def _below_max(fin, q):
    """quantile q of the scores, lowered to the largest value below the maximum when the quantile IS the
    maximum (very narrow PFMs have a handful of distinct scores; `>` the maximum would leave no hit)"""
    if not fin.size:
        return 0.0
    t = float(np.quantile(fin, q))
    if not (fin > t).any():
        lower = fin[fin < fin.max()]
        t = float(lower.max()) if lower.size else float(fin.max()) - 1.0
    return t


@pytest.mark.parametrize("n,m", [(1, 12), (7, 8), (8, 1), (16, 12), (24, 13), (40, 2), (19, 18), (9, 33), (10, 64), (130, 12), (300, 7)])
def test_library_hits_match_oracle_per_motif(ctx, oracle, n, m):
    rng = np.random.default_rng(100 * n + m)
    s = rand_stream(rng, 24, 0, 900, foreign=0.004)
    T, P = make_library(rng, n, m)
    ts, tt = quantile_thresholds(oracle, s, T, P, 0.97, 0.5)
    lib = ctx.library(T, P)
    ctx.stage(s.codes, s.profile)
    got = ctx.library_hits_staged(lib, ts, tt)
    want = oracle_library_hits(oracle, s, T, P, ts, tt)
    assert len(want[0]) > 20
    check(got, want, True)
    info = lib.info()
    assert info["n_motifs"] == n and info["m"] == m and info["passes"] >= 1 and np.isfinite(info["max_prefilter_eps"])
    lib.close()


def test_library_with_minus_inf_cells_and_f64_profile(ctx, oracle):
    """pseudocount-0 PSSMs (-inf log-odds: the nan_to_num path) and float64 profile storage"""
    rng = np.random.default_rng(77)
    s = rand_stream(rng, 30, 5, 700, dtype=np.float64)
    T, P = make_library(rng, 21, 12, inf_frac=0.15)
    ts, tt = quantile_thresholds(oracle, s, T, P, 0.9, 0.3)
    lib = ctx.library(T, P)
    got = ctx.library_hits_host(lib, s.codes, s.profile, ts, tt)
    want = oracle_library_hits(oracle, s, T, P, ts, tt)
    assert len(want[0]) > 20
    check(got, want, True)
    lib.close()


def test_sequence_only_library(ctx, oracle):
    rng = np.random.default_rng(5)
    s = rand_stream(rng, 40, 0, 600)
    T, _ = make_library(rng, 33, 8, struct=False)
    ts, _ = quantile_thresholds(oracle, s, T, None, 0.98, 0.0)
    lib = ctx.library(T, None)
    pos, mo, sq, st = ctx.library_hits_host(lib, s.codes, None, ts)
    want = oracle_library_hits(oracle, s, T, None, ts, None)
    assert st is None and len(want[0]) > 20
    check((pos, mo, sq, st), want, False)
    lib.close()


def test_thresholds_exactly_on_window_scores(ctx, oracle):
    """strict `>`: a threshold equal to the float32 score of existing windows excludes exactly those windows;
    a threshold one float32 ulp below includes them -- the prefilter margin must not lose or invent either"""
    rng = np.random.default_rng(11)
    s = rand_stream(rng, 20, 50, 500, foreign=0.0)
    n, m = 16, 12
    T, P = make_library(rng, n, m)
    ts = np.empty(n)
    for k in range(n):
        sq = oracle.stream_seq(s.codes, T[k])
        fin = np.sort(sq[np.isfinite(sq)])
        ts[k] = float(fin[int(0.98 * fin.size)])                 # ON a score
    tt = np.full(n, -1e30)
    lib = ctx.library(T, P)
    ctx.stage(s.codes, s.profile)
    for thr in (ts, np.nextafter(ts.astype(np.float32), np.float32(-np.inf)).astype(np.float64)):
        got = ctx.library_hits_staged(lib, thr, tt)
        want = oracle_library_hits(oracle, s, T, P, thr, tt)
        check(got, want, True)
    lib.close()


def test_library_equals_single_motif_entry_points(ctx):
    """same hits and the same score bits as pfmscan_hits_host run once per motif"""
    rng = np.random.default_rng(3)
    s = rand_stream(rng, 30, 0, 800)
    n, m = 12, 12
    T, P = make_library(rng, n, m)
    lib = ctx.library(T, P)
    pos, mo, sq, st = ctx.library_hits_host(lib, s.codes, s.profile, 2.0, -20.0)
    assert len(pos) > 50
    for k in range(n):
        motif = ctx.motif(T[k], P[k])
        p1, s1, t1 = ctx.hits_host(motif, s.codes, s.profile, 2.0, -20.0)
        sel = mo == k
        assert np.array_equal(pos[sel], p1)
        assert_f32_bits_equal(sq[sel], s1)
        assert_struct_close(st[sel], t1, tol=1e-9)               # same per-row fp64 arithmetic
        motif.close()
    lib.close()


def test_library_capacity_protocol_and_errors(ctx):
    from rnascan_amd import _lib
    rng = np.random.default_rng(8)
    s = rand_stream(rng, 10, 100, 400)
    T, P = make_library(rng, 8, 8)
    lib = ctx.library(T, P)
    ctx.stage(s.codes, s.profile)
    full = ctx.library_hits_staged(lib, -2.0, -1e30)
    assert len(full[0]) > 100
    with pytest.raises(_lib.CapacityError) as ei:
        ctx.library_hits_staged(lib, -2.0, -1e30, capacity=10)
    assert ei.value.required >= len(full[0])
    again = ctx.library_hits_staged(lib, -2.0, -1e30, capacity=ei.value.required)
    assert np.array_equal(again[0], full[0]) and np.array_equal(again[1], full[1])
    with pytest.raises(ValueError):
        ctx.library_hits_staged(lib, -np.inf, 0.0)                # every window would be a hit
    with pytest.raises(ValueError):
        ctx.library_hits_staged(lib, np.nan, 0.0)
    assert len(ctx.library_hits_staged(lib, np.inf, 0.0)[0]) == 0
    lib.close()
    T7 = T.copy()
    T7[:, :, 4] = 0.0                                             # a 5-letter alphabet is not a library alphabet
    with pytest.raises(ValueError):
        ctx.library(T7, P)


def test_library_with_huge_and_plus_inf_log_odds(ctx, oracle):
    """the integer prefilter has no magnitude limit; a motif with +inf cells (background 0) runs without prefilter"""
    rng = np.random.default_rng(13)
    s = rand_stream(rng, 12, 50, 400)
    T, P = make_library(rng, 9, 8)
    T[0, :, :4] *= 1e4
    T[1, 3, 2] = np.inf
    T[2, 1, 0] = -np.inf
    T[2, 5, 3] = np.inf                                          # +inf and -inf in one motif: NaN scores where both are hit
    ts = np.array([1e4, 5.0, 2.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0])
    tt = np.full(9, -1e30)
    lib = ctx.library(T, P)
    got = ctx.library_hits_host(lib, s.codes, s.profile, ts, tt)
    want = oracle_library_hits(oracle, s, T, P, ts, tt)
    assert len(want[0]) > 20 and np.isinf(want[2]).any()
    check(got, want, True)
    assert np.isinf(lib.info()["max_prefilter_eps"])
    lib.close()


def test_library_hits_dev_unordered_and_overflow(ctx, oracle):
    import torch
    from rnascan_amd import _lib
    rng = np.random.default_rng(21)
    s = rand_stream(rng, 50, 100, 1200)
    n, m = 20, 12
    T, P = make_library(rng, n, m)
    ts, tt = quantile_thresholds(oracle, s, T, P, 0.95, 0.4)
    want = oracle_library_hits(oracle, s, T, P, ts, tt)
    dev = torch.device("cuda", 0)
    codes = torch.from_numpy(s.codes).to(dev)
    prof = torch.from_numpy(s.profile).to(dev)
    lib = ctx.library(T, P)
    nwant = len(want[0])
    for cap in (nwant + 7, max(nwant // 3, 1)):
        hp = torch.empty(cap, dtype=torch.int64, device=dev)
        hm = torch.empty(cap, dtype=torch.int32, device=dev)
        hs = torch.empty(cap, dtype=torch.float32, device=dev)
        ht = torch.empty(cap, dtype=torch.float64, device=dev)
        cnt = torch.full((1,), 12345, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.library_hits_dev(lib, codes.data_ptr(), prof.data_ptr(), _lib.PROFILE_F32, s.n_pos, ts, tt, cap,
                             hp.data_ptr(), hm.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr())
        ctx.synchronize()
        k = int(cnt.item())
        if cap >= nwant:
            assert k == nwant
            order = np.lexsort((hm[:k].cpu().numpy(), hp[:k].cpu().numpy()))
            check((hp[:k].cpu().numpy()[order], hm[:k].cpu().numpy()[order], hs[:k].cpu().numpy()[order],
                   ht[:k].cpu().numpy()[order]), want, True)
        else:
            assert k > cap                                        # incomplete: the count says so
    lib.close()


def test_library_on_tiny_and_empty_streams(ctx, oracle):
    """no window fits / nothing staged / one window: the persistent kernel must leave cleanly"""
    rng = np.random.default_rng(2)
    T, P = make_library(rng, 10, 12)
    lib = ctx.library(T, P)
    from rnascan_amd import pack
    for lengths in ([], [0], [3, 0, 11], [12], [5, 12, 13]):
        codes = [rng.integers(0, 4, size=L).astype(np.uint8) for L in lengths]
        profs = [rng.dirichlet(np.full(7, 0.3), size=L).astype(np.float32) if L else np.zeros((0, 7), np.float32) for L in lengths]
        if not lengths:                                           # an empty stream has no hits (and is no error)
            got = ctx.library_hits_host(lib, np.zeros(0, np.uint8), np.zeros((0, 7), np.float32), -50.0, -1e30)
            assert len(got[0]) == 0
            continue
        s = pack.pack(codes, profs)
        got = ctx.library_hits_host(lib, s.codes, s.profile, -50.0, -1e30)
        want = oracle_library_hits(oracle, s, T, P, np.full(10, -50.0), np.full(10, -1e30))
        assert len(got[0]) == len(want[0]) == 10 * sum(max(L - 11, 0) for L in lengths)
        check(got, want, True)
    lib.close()


def test_sequence_library_beyond_two_to_the_31_positions(ctx, oracle):
    """a stream longer than 2^31 positions goes through several launches (32-bit positions inside a launch, pos_base
    between them): the hits of the last 3 M positions -- across the 2^31 border -- equal the oracle's on that slice, and
    a window planted at the very end is found at its 64-bit position"""
    import torch
    if torch.cuda.mem_get_info()[0] < 30e9:
        pytest.skip("needs 30 GB of free HBM")
    dev = torch.device("cuda", 0)
    n_pos = (1 << 31) + 1500000
    rng = np.random.default_rng(6)
    m, n = 12, 24
    T, _ = make_library(rng, n, m, struct=False)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    codes = torch.randint(0, 4, (n_pos,), dtype=torch.uint8, device=dev, generator=g)
    codes[3000::3001] = 7                                         # separators: records of 3000
    best = np.argmax(T[5, :, :4], axis=1).astype(np.uint8)        # motif 5's best window, planted at the very end
    codes[n_pos - m - 1:n_pos - 1] = torch.from_numpy(best).to(dev)
    codes[n_pos - 1] = 7
    lib = ctx.library(T, None)
    lo = n_pos - 3000000
    tail = codes[lo:].cpu().numpy()
    ts = np.empty(n)
    for j in range(n):                                            # ~1e-4 of the windows per motif
        sq = oracle.stream_seq(tail, T[j]).astype(np.float64)
        ts[j] = np.quantile(sq[np.isfinite(sq)], 1.0 - 1e-4)
    cap = 1 << 24
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hm = torch.empty(cap, dtype=torch.int32, device=dev)
    hs = torch.empty(cap, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    from rnascan_amd import _lib
    ctx.library_hits_dev(lib, codes.data_ptr(), None, _lib.PROFILE_NONE, n_pos, ts, None, cap, hp.data_ptr(), hm.data_ptr(),
                         hs.data_ptr(), None, cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert 0 < k <= cap
    sel = torch.nonzero(hp[:k] >= lo + m).flatten()              # windows wholly inside the slice
    gp, gm, gs = hp[:k][sel].cpu().numpy(), hm[:k][sel].cpu().numpy(), hs[:k][sel].cpu().numpy()
    order = np.lexsort((gm, gp))
    gp, gm, gs = gp[order], gm[order], gs[order]
    wp, wm, ws = [], [], []
    for j in range(n):
        sq = oracle.stream_seq(tail, T[j])
        p = oracle.stream_hits(sq, None, ts[j], -np.inf)
        p = p[p >= m]
        wp.append(p + lo)
        wm.append(np.full(p.size, j, np.int32))
        ws.append(sq[p])
    wp, wm, ws = np.concatenate(wp), np.concatenate(wm), np.concatenate(ws)
    o2 = np.lexsort((wm, wp))
    assert len(gp) == len(wp) > 1000 and (gp > (1 << 31)).any() and (gp < (1 << 31)).any()
    assert np.array_equal(gp, wp[o2]) and np.array_equal(gm, wm[o2])
    assert_f32_bits_equal(gs, ws[o2])
    assert ((gp == n_pos - m - 1) & (gm == 5)).any()
    lib.close()


@pytest.mark.parametrize("kind", ["seq+struct", "seq", "struct"])
def test_library_pipeline_host_equals_staged(ctx, oracle, kind):
    """pfmscan_library_hits_pipeline_host (chunked upload beside the scan, two chunks of device scratch) == stage + scan,
    for all three kinds of library, at chunk sizes from one work segment up; windows in a chunk's overhang are reported by
    the next chunk only"""
    rng = np.random.default_rng({"seq+struct": 1, "seq": 2, "struct": 3}[kind])
    s = rand_stream(rng, 60, 500, 3000, foreign=0.002)
    assert s.n_pos > 5 * 16384
    T, P = make_library(rng, 14, 11)
    if kind == "seq":
        P = None
    if kind == "struct":
        T = None
    lib = ctx.library(T, P)
    thr_s = 3.0 if T is not None else None
    thr_t = -6.0 if P is not None else None
    want = ctx.library_hits_host(lib, s.codes if T is not None else None, s.profile if P is not None else None, thr_s, thr_t)
    assert len(want[0]) > 200
    for chunk in (1, 16384, 40000, 1 << 24):
        got = ctx.library_hits_pipeline_host(lib, s.codes if T is not None else None, s.profile if P is not None else None, thr_s, thr_t,
                                             chunk_positions=chunk)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        if T is not None:
            assert_f32_bits_equal(got[2], want[2])
        if P is not None:
            assert np.array_equal(got[3].view(np.uint64), want[3].view(np.uint64))
    with pytest.raises(ValueError):
        ctx.scan_staged(ctx.motif(np.full((3, 8), np.nan)))            # nothing stays staged
    lib.close()
