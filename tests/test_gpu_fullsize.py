"""BASELINE.json's full C3 size (100k records x 3 kb, seq + struct, w = 12) on the device,
checked through size-independent properties plus an oracle comparison of sampled records:
  * doubling both PSSMs doubles every score EXACTLY (x2 is exact in fp64 and in the f32 cast)
  * a launch is deterministic (two runs are bit-identical)
  * hits mode finds exactly the windows the all-scores output says pass the thresholds
  * 64 records picked at random score the same inside the 300M-position stream as on their
    own through the host API, and match the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R, L, M = 100000, 3000, 12


@pytest.fixture(scope="module")
def big():
    import torch
    import bench
    from rnascan_amd import _lib
    if torch.cuda.mem_get_info()[0] < 40e9:
        pytest.skip("needs 40 GB of free HBM")
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    table, spssm = bench.make_pssms(M)
    codes, profile, n_pos = bench.make_stream(torch, dev, R, L, 424242)
    torch.cuda.synchronize()
    yield dict(torch=torch, dev=dev, ctx=ctx, table=table, spssm=spssm, codes=codes, profile=profile, n_pos=n_pos)
    ctx.close()


def _scan(b, table, spssm):
    torch, _lib = b["torch"], __import__("rnascan_amd._lib", fromlist=["x"])
    out_seq = torch.empty(b["n_pos"], dtype=torch.float32, device=b["dev"])
    out_st = torch.empty(b["n_pos"], dtype=torch.float64, device=b["dev"])
    motif = b["ctx"].motif(table, spssm)
    b["ctx"].scan_dev(motif, b["codes"].data_ptr(), b["profile"].data_ptr(), _lib.PROFILE_F32, b["n_pos"],
                      out_seq.data_ptr(), out_st.data_ptr())
    b["ctx"].synchronize()
    motif.close()
    return out_seq, out_st


def test_fullsize_properties(big, oracle):
    torch = big["torch"]
    from rnascan_amd import _lib, pack
    s1, t1 = _scan(big, big["table"], big["spssm"])
    s1b, t1b = _scan(big, big["table"], big["spssm"])
    assert torch.equal(s1.view(torch.int32), s1b.view(torch.int32)) and torch.equal(t1.view(torch.int64), t1b.view(torch.int64))
    del s1b, t1b
    # windows: every record contributes L-M+1 finite scores, the rest (separator-touching) is NaN
    assert int(torch.isnan(s1).sum()) == R * M and not bool(torch.isnan(t1[: big["n_pos"] - M]).any())
    s2, t2 = _scan(big, 2.0 * big["table"], 2.0 * big["spssm"])
    ok = ~torch.isnan(s1)
    assert torch.equal(s2[ok], 2.0 * s1[ok])
    assert torch.equal(t2[: big["n_pos"] - M], 2.0 * t1[: big["n_pos"] - M])
    del s2, t2
    # hits mode == thresholding the all-scores output
    thr_s, thr_t = 2.0, -12.0
    want = torch.nonzero((s1.double() > thr_s) & (t1 > thr_t)).flatten()
    cap = int(want.numel()) + 16
    hp = torch.empty(cap, dtype=torch.int64, device=big["dev"])
    hs = torch.empty(cap, dtype=torch.float32, device=big["dev"])
    ht = torch.empty(cap, dtype=torch.float64, device=big["dev"])
    cnt = torch.zeros(1, dtype=torch.int64, device=big["dev"])
    motif = big["ctx"].motif(big["table"], big["spssm"])
    torch.cuda.synchronize()                     # the zero fill runs on torch's stream, the scan on the ctx's own
    big["ctx"].hits_dev(motif, big["codes"].data_ptr(), big["profile"].data_ptr(), _lib.PROFILE_F32, big["n_pos"],
                        thr_s, thr_t, cap, hp.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr())
    big["ctx"].synchronize()
    k = int(cnt.item())
    assert k == int(want.numel()) and k > 100
    order = torch.argsort(hp[:k])
    assert torch.equal(hp[:k][order], want)
    assert torch.equal(hs[:k][order], s1[want]) and torch.equal(ht[:k][order], t1[want])
    # letters-only hits (k_letters_pre: fp32 prefilter, 32 tiles per workgroup, LDS hit queues) == thresholding
    # the all-scores output, at a selective and at a dense threshold
    lo_motif = big["ctx"].motif(letter_table=big["table"])
    for thr in (6.0, -1.0):
        want = torch.nonzero(s1.double() > thr).flatten()
        cap = int(want.numel()) + 16
        hp = torch.empty(cap, dtype=torch.int64, device=big["dev"])
        hs = torch.empty(cap, dtype=torch.float32, device=big["dev"])
        cnt = torch.zeros(1, dtype=torch.int64, device=big["dev"])
        torch.cuda.synchronize()                 # the zero fill runs on torch's stream, the scan on the ctx's own
        big["ctx"].hits_dev(lo_motif, big["codes"].data_ptr(), None, _lib.PROFILE_NONE, big["n_pos"],
                            thr, -np.inf, cap, hp.data_ptr(), hs.data_ptr(), None, cnt.data_ptr())
        big["ctx"].synchronize()
        k = int(cnt.item())
        assert k == int(want.numel()) and k > 100
        order = torch.argsort(hp[:k])
        assert torch.equal(hp[:k][order], want) and torch.equal(hs[:k][order], s1[want])
        del hp, hs, want, order
    lo_motif.close()
    # the candidate-then-verify combined scan finds the same hits as thresholding both outputs
    thr_s, thr_t = 5.0, -14.0
    want = torch.nonzero((s1.double() > thr_s) & (t1 > thr_t)).flatten()
    cap = int(want.numel()) + 16
    hp = torch.empty(cap, dtype=torch.int64, device=big["dev"])
    hs = torch.empty(cap, dtype=torch.float32, device=big["dev"])
    ht = torch.empty(cap, dtype=torch.float64, device=big["dev"])
    cnt = torch.zeros(1, dtype=torch.int64, device=big["dev"])
    torch.cuda.synchronize()
    big["ctx"].hits_adaptive_dev(motif, big["codes"].data_ptr(), big["profile"].data_ptr(), _lib.PROFILE_F32, big["n_pos"],
                                 thr_s, thr_t, cap, hp.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr())
    big["ctx"].synchronize()
    k = int(cnt.item())
    assert k == int(want.numel()) and k > 100
    order = torch.argsort(hp[:k])
    assert torch.equal(hp[:k][order], want) and torch.equal(hs[:k][order], s1[want])
    assert float((ht[:k][order] - t1[want]).abs().max()) <= 1e-6      # per-row exact path vs fused chain
    del hp, hs, ht, want, order
    # sampled records: stream position independence + oracle
    rng = np.random.default_rng(0)
    stride = L + 1
    for r in rng.choice(R, size=64, replace=False):
        lo = int(r) * stride
        c = big["codes"][lo:lo + stride].cpu().numpy()
        p = big["profile"][lo:lo + stride].cpu().numpy()
        sq, st = big["ctx"].scan_host(motif, c, p)
        assert np.array_equal(sq[: L - M + 1].view(np.uint32), s1[lo:lo + L - M + 1].cpu().numpy().view(np.uint32))
        assert np.array_equal(st[: L - M + 1], t1[lo:lo + L - M + 1].cpu().numpy())
        ref_sq = oracle.stream_seq(c, big["table"])
        ref_st = oracle.stream_struct(p, big["spssm"])
        assert np.array_equal(sq[: L - M + 1].view(np.uint32), ref_sq[: L - M + 1].view(np.uint32))
        assert np.abs(st[: L - M + 1] - ref_st[: L - M + 1]).max() <= 1e-6
    motif.close()


def test_fullsize_c2_sequence_only(big, oracle):
    """BASELINE configs[1] at full size: 100k x 3 kb, sequence PFM of width 8, all-scores through
    k_letters<5, float, false, 8> (73k workgroups): deterministic, x2 exact, NaN exactly at the separator-touching
    windows, 64 sampled records equal to the oracle bit for bit"""
    import bench
    torch = big["torch"]
    from rnascan_amd import _lib
    m = 8
    table, _ = bench.make_pssms(m)
    ctx, codes, n_pos, dev = big["ctx"], big["codes"], big["n_pos"], big["dev"]

    def scan(tab):
        out = torch.empty(n_pos, dtype=torch.float32, device=dev)
        motif = ctx.motif(letter_table=tab)
        ctx.scan_dev(motif, codes.data_ptr(), None, _lib.PROFILE_NONE, n_pos, out.data_ptr(), None)
        ctx.synchronize()
        motif.close()
        return out

    s1, s1b = scan(table), scan(table)
    assert torch.equal(s1.view(torch.int32), s1b.view(torch.int32))
    del s1b
    assert int(torch.isnan(s1).sum()) == R * m                  # the m windows per record that touch its separator
    s2 = scan(2.0 * table)
    ok = ~torch.isnan(s1)
    assert torch.equal(s2[ok], 2.0 * s1[ok])
    del s2, ok
    rng = np.random.default_rng(1)
    stride = L + 1
    for r in rng.choice(R, size=64, replace=False):
        lo = int(r) * stride
        c = codes[lo:lo + stride].cpu().numpy()
        ref = oracle.stream_seq(c, table)
        got = s1[lo:lo + stride].cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        v = ~np.isnan(ref)
        assert np.array_equal(got[v].view(np.uint32), ref[v].view(np.uint32))
    # hits mode over the same stream (k_letters_pre) == thresholding the all-scores output
    motif = ctx.motif(letter_table=table)
    want = torch.nonzero(s1.double() > 6.0).flatten()
    cap = int(want.numel()) + 16
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hs = torch.empty(cap, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.hits_dev(motif, codes.data_ptr(), None, _lib.PROFILE_NONE, n_pos, 6.0, -np.inf, cap, hp.data_ptr(), hs.data_ptr(), None, cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert k == int(want.numel()) and k > 1000
    order = torch.argsort(hp[:k])
    assert torch.equal(hp[:k][order], want) and torch.equal(hs[:k][order], s1[want])
    motif.close()


def test_fullsize_c5_library_equals_all_scores(big, oracle):
    """BASELINE configs[4] at full size: 256 seq+struct PFM pairs over the 100k x 3 kb stream through the one-pass library
    kernel (three passes of 96 / 96 / 64 motifs).  For motifs of every pass the library's hits are exactly the windows
    the all-scores kernel puts above both thresholds: same positions, same float32 sequence scores, structure scores
    within 1e-6; and the total matches the sum over ALL motifs of a thresholded count (no motif is skipped)"""
    import bench
    torch = big["torch"]
    from rnascan_amd import _lib
    ctx, codes, profile, n_pos, dev = big["ctx"], big["codes"], big["profile"], big["n_pos"], big["dev"]
    n = 256
    tabs = [bench.make_pssms(M, "finite", seed=1000 + k) for k in range(n)]
    lib = ctx.library(np.stack([t for t, _ in tabs]), np.stack([p for _, p in tabs]))
    thr_s, thr_t = 6.0, -5.03125
    cap = 1 << 25
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hm = torch.empty(cap, dtype=torch.int32, device=dev)
    hs = torch.empty(cap, dtype=torch.float32, device=dev)
    ht = torch.empty(cap, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.library_hits_dev(lib, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, thr_s, thr_t, cap,
                         hp.data_ptr(), hm.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert 1000000 < k <= cap
    assert lib.info()["passes"] == 3
    per_motif = torch.bincount(hm[:k].long(), minlength=n)
    assert int(per_motif.min()) > 0                              # every motif of every pass reports hits
    out_seq = torch.empty(n_pos, dtype=torch.float32, device=dev)
    out_st = torch.empty(n_pos, dtype=torch.float64, device=dev)
    for mk in (0, 95, 96, 191, 192, 255):                         # first / last motif of each pass
        motif = ctx.motif(*tabs[mk])
        ctx.scan_dev(motif, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, out_seq.data_ptr(), out_st.data_ptr())
        ctx.synchronize()
        motif.close()
        want = torch.nonzero((out_seq.double() > thr_s) & (out_st > thr_t)).flatten()
        sel = torch.nonzero(hm[:k] == mk).flatten()
        assert int(sel.numel()) == int(want.numel()) == int(per_motif[mk])
        order = torch.argsort(hp[:k][sel])
        idx = sel[order]
        assert torch.equal(hp[:k][idx], want)
        assert torch.equal(hs[:k][idx].view(torch.int32), out_seq[want].view(torch.int32))
        assert float((ht[:k][idx] - out_st[want]).abs().max()) <= 1e-6
    # the three passes ran side by side as teams of one launch (a stream this long); one after the other: the same hits
    os.environ["PFMSCAN_LIB_SEQUENTIAL"] = "1"
    try:
        hm2 = torch.empty(cap, dtype=torch.int32, device=dev)
        hp2 = torch.empty(cap, dtype=torch.int64, device=dev)
        cnt.zero_()
        torch.cuda.synchronize()
        ctx.library_hits_dev(lib, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, thr_s, thr_t, cap,
                             hp2.data_ptr(), hm2.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr())
        ctx.synchronize()
    finally:
        del os.environ["PFMSCAN_LIB_SEQUENTIAL"]
    assert int(cnt.item()) == k
    assert torch.equal(torch.bincount(hm2[:k].long(), minlength=n), per_motif)
    key = lambda pos, mot: torch.sort(pos * n + mot.long()).values
    assert torch.equal(key(hp[:k], hm[:k]), key(hp2[:k], hm2[:k]))
    lib.close()


def test_fullsize_struct_only_library_equals_all_scores(big):
    """the structure side of BASELINE configs[4] alone at full size: 64 structure PFMs over the 100k x 3 kb profile in ONE
    pass of k_profile_lib (every fourth motif with -inf cells: both forms of the kernel).  For motifs of every group
    the library's hits are exactly the windows the single-motif all-scores kernel puts above the threshold, with
    bit-identical scores; the per-motif counts add up to the total"""
    import bench
    torch = big["torch"]
    from rnascan_amd import _lib
    ctx, profile, n_pos, dev = big["ctx"], big["profile"], big["n_pos"], big["dev"]
    n = 64
    P = np.stack([bench.make_pssms(M, "inf" if k % 4 == 3 else "finite", seed=1000 + k)[1] for k in range(n)])
    lib = ctx.library(None, P)
    thr = np.where(np.arange(n) % 4 == 3, 0.5, 2.5)
    cap = 1 << 26
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hm = torch.empty(cap, dtype=torch.int32, device=dev)
    ht = torch.empty(cap, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.library_hits_dev(lib, None, profile.data_ptr(), _lib.PROFILE_F32, n_pos, None, thr, cap, hp.data_ptr(), hm.data_ptr(), None,
                         ht.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert 10000 < k <= cap
    per_motif = torch.bincount(hm[:k].long(), minlength=n)
    out_st = torch.empty(n_pos, dtype=torch.float64, device=dev)
    for mk in (0, 1, 2, 3, 31, 62, 63):
        motif = ctx.motif(None, P[mk])
        ctx.scan_dev(motif, None, profile.data_ptr(), _lib.PROFILE_F32, n_pos, None, out_st.data_ptr())
        ctx.synchronize()
        motif.close()
        want = torch.nonzero(out_st > float(thr[mk])).flatten()
        sel = torch.nonzero(hm[:k] == mk).flatten()
        assert int(sel.numel()) == int(want.numel()) == int(per_motif[mk])
        order = torch.argsort(hp[:k][sel])
        idx = sel[order]
        assert torch.equal(hp[:k][idx], want)
        assert torch.equal(ht[:k][idx].view(torch.int64), out_st[want].view(torch.int64))       # bit-identical to k_profile
    assert int(per_motif.sum()) == k
    lib.close()


def test_engine_takes_the_library_pipeline_for_long_host_streams():
    """a host-resident stream longer than 2 x 2^24 positions that only one library scans: HipEngine.library_hits goes
    through pfmscan_library_hits_pipeline_host (three chunks here, upload beside scan) and returns what stage + scan returns"""
    import torch
    import bench
    from rnascan_amd import pack, scanner
    dev = torch.device("cuda", 0)
    R2, L2 = 12500, 3000
    codes, profile, n_pos = bench.make_stream(torch, dev, R2, L2, 99)
    assert n_pos > scanner.PIPELINE_MIN
    lengths = np.full(R2, L2, dtype=np.int64)
    offsets = np.arange(R2, dtype=np.int64) * (L2 + 1)
    stream = pack.Stream(codes.cpu().numpy(), profile.cpu().numpy(), offsets, lengths)
    del codes, profile
    tabs = [bench.make_pssms(M, "finite", seed=1000 + k) for k in range(24)]
    T, P = np.stack([t for t, _ in tabs]), np.stack([p for _, p in tabs])
    eng = scanner.HipEngine(0)
    calls = []
    real = eng.ctx.library_hits_pipeline_host
    eng.ctx.library_hits_pipeline_host = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    got = eng.library_hits(stream, T, P, 6.0, -6.0)                       # one_shot: the pipeline
    assert calls == [1]
    want = eng.library_hits(stream, T, P, 6.0, -6.0, one_shot=False)      # staged once, scanned from the device
    assert calls == [1]
    assert len(want[0]) > 1000
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[2].view(np.uint32), want[2].view(np.uint32))
    assert np.array_equal(got[3].view(np.uint64), want[3].view(np.uint64))
    # structure-only library over the same profile (a fresh Stream object: the one above is staged by now)
    stream = pack.Stream(stream.codes, stream.profile, offsets, lengths)
    got = eng.library_hits(stream, None, P, None, 2.0)
    assert calls == [1, 1]
    want = eng.library_hits(stream, None, P, None, 2.0, one_shot=False)
    assert len(want[0]) > 1000
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal(got[3].view(np.uint64), want[3].view(np.uint64))
    eng.close()


def test_fullsize_c4_shard_of_125k_records(oracle):
    """one shard of BASELINE configs[3] (1M records over 8 GPUs = 125k x 3 kb per GPU, 375M positions): deterministic,
    NaN pattern, x2 exactness, hits == thresholded scores, 32 sampled records against the oracle"""
    import torch
    import bench
    from rnascan_amd import _lib
    if torch.cuda.mem_get_info()[0] < 60e9:
        pytest.skip("needs 60 GB of free HBM")
    dev = torch.device("cuda", 0)
    R4 = 125000
    ctx = _lib.Context(0)
    table, spssm = bench.make_pssms(M)
    codes, profile, n_pos = bench.make_stream(torch, dev, R4, L, 20240601 + 3)      # rank 3's stream of the bench
    torch.cuda.synchronize()
    motif = ctx.motif(table, spssm)

    def scan(mo):
        a = torch.empty(n_pos, dtype=torch.float32, device=dev)
        b = torch.empty(n_pos, dtype=torch.float64, device=dev)
        ctx.scan_dev(mo, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, a.data_ptr(), b.data_ptr())
        ctx.synchronize()
        return a, b

    s1, t1 = scan(motif)
    s1b, t1b = scan(motif)
    assert torch.equal(s1.view(torch.int32), s1b.view(torch.int32)) and torch.equal(t1.view(torch.int64), t1b.view(torch.int64))
    del s1b, t1b
    assert int(torch.isnan(s1).sum()) == R4 * M
    m2 = ctx.motif(2.0 * table, 2.0 * spssm)
    s2, t2 = scan(m2)
    m2.close()
    ok = ~torch.isnan(s1)
    assert torch.equal(s2[ok], 2.0 * s1[ok]) and torch.equal(t2[: n_pos - M], 2.0 * t1[: n_pos - M])
    del s2, t2, ok
    thr_s, thr_t = 5.0, -9.0
    want = torch.nonzero((s1.double() > thr_s) & (t1 > thr_t)).flatten()
    cap = int(want.numel()) + 16
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hs = torch.empty(cap, dtype=torch.float32, device=dev)
    ht = torch.empty(cap, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.hits_adaptive_dev(motif, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, thr_s, thr_t, cap,
                          hp.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert k == int(want.numel()) and k > 100
    order = torch.argsort(hp[:k])
    assert torch.equal(hp[:k][order], want) and torch.equal(hs[:k][order], s1[want])
    assert float((ht[:k][order] - t1[want]).abs().max()) <= 1e-6
    rng = np.random.default_rng(4)
    stride = L + 1
    for r in rng.choice(R4, size=32, replace=False):
        lo = int(r) * stride
        c = codes[lo:lo + stride].cpu().numpy()
        p = profile[lo:lo + stride].cpu().numpy()
        ref_sq, ref_st = oracle.stream_seq(c, table), oracle.stream_struct(p, spssm)
        got_sq, got_st = s1[lo:lo + L - M + 1].cpu().numpy(), t1[lo:lo + L - M + 1].cpu().numpy()
        assert np.array_equal(got_sq.view(np.uint32), ref_sq[: L - M + 1].view(np.uint32))
        assert np.abs(got_st - ref_st[: L - M + 1]).max() <= 1e-6
    motif.close()
    ctx.close()


def test_fullsize_c4_whole_on_one_gpu(oracle):
    """ALL of BASELINE configs[3] -- 1M records x 3 kb = 3.0x10^9 positions (past 2^31), 87 GB of codes + profile in, 36 GB of
    scores out -- resident on ONE MI355X and scanned by single launches: deterministic, NaN pattern, fused hits == thresholded
    scores at 64-bit positions, 48 sampled records (half of them past position 2^31) against the oracle"""
    import torch
    import bench
    from rnascan_amd import _lib
    if torch.cuda.mem_get_info()[0] < 200e9:
        pytest.skip("needs 200 GB of free HBM")
    dev = torch.device("cuda", 0)
    R4 = 1000000
    ctx = _lib.Context(0)
    table, spssm = bench.make_pssms(M)
    codes, profile, n_pos = bench.make_stream(torch, dev, R4, L, 20240601)
    torch.cuda.synchronize()
    assert n_pos > (1 << 31)
    motif = ctx.motif(table, spssm)

    def scan():
        a = torch.empty(n_pos, dtype=torch.float32, device=dev)
        b = torch.empty(n_pos, dtype=torch.float64, device=dev)
        ctx.scan_dev(motif, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, a.data_ptr(), b.data_ptr())
        ctx.synchronize()
        return a, b

    s1, t1 = scan()
    s1b, t1b = scan()
    assert torch.equal(s1.view(torch.int32), s1b.view(torch.int32)) and torch.equal(t1.view(torch.int64), t1b.view(torch.int64))
    del s1b, t1b
    torch.cuda.empty_cache()
    assert sum(int(torch.isnan(s1[lo:lo + (1 << 30)]).sum()) for lo in range(0, n_pos, 1 << 30)) == R4 * M
    thr_s, thr_t = 6.0, -8.0
    step = 1 << 30                                                   # (torch.nonzero does not take 2^31 elements at once)
    want = torch.cat([torch.nonzero((s1[lo:lo + step].double() > thr_s) & (t1[lo:lo + step] > thr_t)).flatten() + lo
                      for lo in range(0, n_pos, step)])
    cap = int(want.numel()) + 16
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hs = torch.empty(cap, dtype=torch.float32, device=dev)
    ht = torch.empty(cap, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.hits_adaptive_dev(motif, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, thr_s, thr_t, cap,
                          hp.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert k == int(want.numel()) and k > 1000
    order = torch.argsort(hp[:k])
    assert torch.equal(hp[:k][order], want) and torch.equal(hs[:k][order], s1[want])
    assert float((ht[:k][order] - t1[want]).abs().max()) <= 1e-6
    assert int(want.max()) > (1 << 31)
    rng = np.random.default_rng(44)
    stride = L + 1
    first_past = ((1 << 31) + stride - 1) // stride
    picks = np.concatenate([rng.choice(first_past, size=24, replace=False), first_past + rng.choice(R4 - first_past, size=24, replace=False)])
    for r in picks:
        lo = int(r) * stride
        c = codes[lo:lo + stride].cpu().numpy()
        p = profile[lo:lo + stride].cpu().numpy()
        ref_sq, ref_st = oracle.stream_seq(c, table), oracle.stream_struct(p, spssm)
        got_sq, got_st = s1[lo:lo + L - M + 1].cpu().numpy(), t1[lo:lo + L - M + 1].cpu().numpy()
        assert np.array_equal(got_sq.view(np.uint32), ref_sq[: L - M + 1].view(np.uint32))
        assert np.abs(got_st - ref_st[: L - M + 1]).max() <= 1e-6
    motif.close()
    ctx.close()
    del s1, t1, codes, profile, hp, hs, ht, want
    torch.cuda.empty_cache()


def test_fullsize_structure_letter_strings_and_two_fasta(big, oracle):
    """SURVEY 8f N4 at full size (100k x 3 kb structure strings, 7 letters, w = 12): the fp64 letter hits are exactly the
    windows the all-scores output (k_letters<..., double>) says exceed the threshold, with the same bits; doubling the
    table doubles every hit's score exactly and keeps the hit set at the doubled threshold; the two-stream hits are the
    intersection of the two single-stream hit sets (combine()'s inner join, rnascan.py:416-434); 32 records picked at random
    match the CPU oracle."""
    torch = big["torch"]
    from rnascan_amd import _lib
    ctx, dev, n_pos = big["ctx"], big["dev"], big["n_pos"]
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    scodes = torch.randint(0, 7, (R, L + 1), dtype=torch.uint8, device=dev, generator=g)
    scodes |= (torch.rand((R, L + 1), device=dev, generator=g) < 0.25).to(torch.uint8) * 8        # letters written in lower case
    scodes[torch.rand((R, L + 1), device=dev, generator=g) < 0.0005] = 7                           # foreign letters
    scodes[:, L] = 7
    scodes = scodes.view(-1)
    rng = np.random.default_rng(5)
    T = np.full((M, 8), np.nan)
    T[:, :7] = rng.normal(-0.6, 1.8, size=(M, 7))
    T[rng.integers(0, M), rng.integers(0, 7)] = -np.inf
    mo = ctx.motif(T, None)
    full = torch.empty(n_pos, dtype=torch.float64, device=dev)
    ctx.scan_letters_f64_dev(mo, scodes.data_ptr(), n_pos, full.data_ptr())
    ctx.synchronize()
    fin = full[torch.isfinite(full)]
    thr = float(torch.quantile(fin[:4_000_000], 0.9995)) + 1e-7
    cap = 1 << 22
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hs = torch.empty(cap, dtype=torch.float32, device=dev)
    ht = torch.empty(cap, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def letter_hits(motif, threshold):
        cnt.zero_()
        torch.cuda.synchronize()
        ctx.hits_letters_f64_dev(motif, scodes.data_ptr(), n_pos, threshold, cap, hp.data_ptr(), ht.data_ptr(), cnt.data_ptr())
        ctx.synchronize()
        k = int(cnt.item())
        assert 0 < k <= cap
        order = torch.argsort(hp[:k])
        return hp[:k][order].clone(), ht[:k][order].clone()

    pos, sc = letter_hits(mo, thr)
    want = torch.nonzero(full > thr).flatten()
    assert torch.equal(pos, want) and torch.equal(sc.view(torch.int64), full[want].view(torch.int64))
    assert 50_000 < pos.numel() < 400_000
    mo2 = ctx.motif(2.0 * T, None)
    pos2, sc2 = letter_hits(mo2, 2.0 * thr)
    assert torch.equal(pos2, pos) and torch.equal(sc2, 2.0 * sc)
    mo2.close()
    # two code streams: sequence PFM (float32 compare) AND structure letters (fp64 compare)
    mq = ctx.motif(big["table"], None)
    sq = torch.empty(n_pos, dtype=torch.float32, device=dev)
    ctx.scan_dev(mq, big["codes"].data_ptr(), None, _lib.PROFILE_NONE, n_pos, sq.data_ptr(), None)
    ctx.synchronize()
    thr_q, thr_t = 4.0, float(torch.quantile(fin[:4_000_000], 0.9)) + 1e-7
    cnt.zero_()
    torch.cuda.synchronize()
    ctx.hits_pair_dev(mq, mo, big["codes"].data_ptr(), scodes.data_ptr(), n_pos, thr_q, thr_t, cap, hp.data_ptr(), hs.data_ptr(),
                      ht.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    order = torch.argsort(hp[:k])
    want = torch.nonzero((sq.double() > thr_q) & (full > thr_t)).flatten()
    assert k == want.numel() and k > 1000 and torch.equal(hp[:k][order], want)
    assert torch.equal(hs[:k][order].view(torch.int32), sq[want].view(torch.int32))
    assert torch.equal(ht[:k][order].view(torch.int64), full[want].view(torch.int64))
    # sampled records against the CPU oracle
    stride = L + 1
    for r in rng.integers(0, R, size=32).tolist():
        rec = scodes[r * stride:(r + 1) * stride].cpu().numpy()
        w = oracle.stream_letters_f64(rec, T)
        gq = full[r * stride:(r + 1) * stride].cpu().numpy()
        ok = ~np.isnan(w)
        assert np.array_equal(np.isnan(gq), ~ok) and np.array_equal(gq[ok], w[ok])
    mo.close()
    mq.close()
