"""Single-motif scans of a stream LONGER THAN 2^31 positions (BASELINE config 4 is 3x10^9 positions; the product shards it,
but the entry points take an int64 n_pos and must not wrap): the scores and hits of the last 3 M positions -- across the 2^31
border -- equal the oracle's on that slice (window scores do not depend on what lies before the window: _pwm.c:34-68,
matrix.py:25-43), and a window planted at the very end is found at its 64-bit position.  Letters side only: a profile of
that length is 60 GB.  The library form of this test is in test_gpu_library.py."""
import numpy as np
import pytest

from conftest import assert_f32_bits_equal
from test_gpu_parity import rand_table

pytestmark = pytest.mark.gpu

N_POS = (1 << 31) + 1500000
TAIL = 3000000


@pytest.fixture(scope="module")
def big(ctx):
    import torch
    if torch.cuda.mem_get_info()[0] < 40e9:
        pytest.skip("needs 40 GB of free HBM")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(123)
    codes = torch.randint(0, 4, (N_POS,), dtype=torch.uint8, device=dev, generator=g)
    codes[3000::3001] = 7                                         # separators: records of 3000
    codes[N_POS - 1] = 7
    yield codes
    del codes
    torch.cuda.empty_cache()


def _plant(codes, T, m):
    import torch
    best = np.argmax(np.where(np.isfinite(T[:, :4]), T[:, :4], -1e300), axis=1).astype(np.uint8)
    codes[N_POS - m - 1:N_POS - 1] = torch.from_numpy(best).to(codes.device)


@pytest.mark.parametrize("m", [8, 12, 20, 40])                   # k_letters_fixed (2..32) and the generic k_letters
def test_all_scores_beyond_two_to_the_31_positions(ctx, oracle, big, m):
    import torch
    from rnascan_amd import _lib
    rng = np.random.default_rng(40 + m)
    T = rand_table(rng, m)
    motif = ctx.motif(letter_table=T)
    out = torch.empty(N_POS, dtype=torch.float32, device=big.device)
    torch.cuda.synchronize()
    ctx.scan_dev(motif, big.data_ptr(), None, _lib.PROFILE_NONE, N_POS, out.data_ptr(), None)
    ctx.synchronize()
    lo = N_POS - TAIL
    want = oracle.stream_seq(big[lo:].cpu().numpy(), T)
    assert_f32_bits_equal(out[lo:].cpu().numpy(), want)
    head = oracle.stream_seq(big[:TAIL].cpu().numpy(), T)          # ... and the first 3 M (the last m - 1 of them see what follows)
    assert_f32_bits_equal(out[:TAIL - m].cpu().numpy(), head[:TAIL - m])
    mid = (1 << 31) - 1000                                          # the border itself
    wmid = oracle.stream_seq(big[mid:mid + 5000].cpu().numpy(), T)
    assert_f32_bits_equal(out[mid:mid + 5000 - m].cpu().numpy(), wmid[:5000 - m])
    motif.close()
    del out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("m,q", [(6, 1e-3), (12, 1e-4), (18, 1e-4), (30, 1e-5)])     # the credit prefilters of every width bucket
def test_hits_beyond_two_to_the_31_positions(ctx, oracle, big, m, q):
    import torch
    from rnascan_amd import _lib
    rng = np.random.default_rng(70 + m)
    T = rand_table(rng, m)
    _plant(big, T, m)
    motif = ctx.motif(letter_table=T)
    lo = N_POS - TAIL
    tail = big[lo:].cpu().numpy()
    sq = oracle.stream_seq(tail, T)
    thr = float(np.quantile(sq[np.isfinite(sq)].astype(np.float64), 1.0 - q))
    cap = 1 << 24
    hp = torch.empty(cap, dtype=torch.int64, device=big.device)
    hs = torch.empty(cap, dtype=torch.float32, device=big.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=big.device)
    torch.cuda.synchronize()
    ctx.hits_dev(motif, big.data_ptr(), None, _lib.PROFILE_NONE, N_POS, thr, -np.inf, cap, hp.data_ptr(), hs.data_ptr(), None,
                 cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert 0 < k <= cap, k
    gp, gs = hp[:k].cpu().numpy(), hs[:k].cpu().numpy()
    order = np.argsort(gp, kind="stable")
    gp, gs = gp[order], gs[order]
    assert np.unique(gp).size == gp.size and gp.min() >= 0 and gp.max() < N_POS
    sel = gp >= lo + m                                             # windows wholly inside the slice
    wp = oracle.stream_hits(sq, None, thr, -np.inf)
    wp = wp[wp >= m]
    assert np.array_equal(gp[sel], wp + lo), (m, int(sel.sum()), wp.size)
    assert_f32_bits_equal(gs[sel], sq[wp])
    assert (gp == N_POS - m - 1).any() and (gp < (1 << 31)).any() and (gp > (1 << 31)).any()
    # the border itself
    mid = (1 << 31) - 100000
    smid = oracle.stream_seq(big[mid:mid + 200000].cpu().numpy(), T)
    wmid = oracle.stream_hits(smid, None, thr, -np.inf)
    wmid = wmid[wmid < 200000 - m] + mid
    got_mid = gp[(gp >= mid) & (gp < mid + 200000 - m)]
    assert np.array_equal(got_mid, wmid)
    motif.close()


def test_fp64_letter_hits_beyond_two_to_the_31_positions(ctx, oracle, big):
    """the generic-alphabet route (matrix.py:25-43: fp64 sum, no float32 cast): k_letters_cred8 + its verify pass"""
    import torch
    m = 12
    rng = np.random.default_rng(5)
    T = rand_table(rng, m)
    _plant(big, T, m)
    motif = ctx.motif(letter_table=T)
    lo = N_POS - TAIL
    full = oracle.stream_letters_f64(big[lo:].cpu().numpy(), T)
    thr = float(np.quantile(full[np.isfinite(full)], 1.0 - 1e-4))
    cap = 1 << 24
    hp = torch.empty(cap, dtype=torch.int64, device=big.device)
    hv = torch.empty(cap, dtype=torch.float64, device=big.device)
    cnt = torch.zeros(1, dtype=torch.int64, device=big.device)
    torch.cuda.synchronize()
    ctx.hits_letters_f64_dev(motif, big.data_ptr(), N_POS, thr, cap, hp.data_ptr(), hv.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    assert 0 < k <= cap, k
    gp, gv = hp[:k].cpu().numpy(), hv[:k].cpu().numpy()
    order = np.argsort(gp, kind="stable")
    gp, gv = gp[order], gv[order]
    sel = gp >= lo + m
    wp = np.flatnonzero(full > thr)
    wp = wp[wp >= m]
    assert np.array_equal(gp[sel], wp + lo)
    assert np.array_equal(gv[sel], full[wp])
    assert (gp == N_POS - m - 1).any() and (gp < (1 << 31)).any()
    motif.close()
