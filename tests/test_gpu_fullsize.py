"""BASELINE.json's full C3 size (100k records x 3 kb, seq + struct, w = 12) on the device,
checked through size-independent properties plus an oracle comparison of sampled records:
  * doubling both PSSMs doubles every score EXACTLY (x2 is exact in fp64 and in the f32 cast)
  * a launch is deterministic (two runs are bit-identical)
  * hits mode finds exactly the windows the all-scores output says pass the thresholds
  * 64 records picked at random score the same inside the 300M-position stream as on their
    own through the host API, and match the CPU oracle."""
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
