"""PFMs wider than PFMSCAN_MAX_M = 64 (the reference's loops take any width: _pwm.c:34-68, matrix.py:25-43,
rnascan.py:302-307).  Letters-only scans run the slab kernel (k_wide_letters), scans with a structure part the plain
one-thread-per-window kernel (k_wide); every mode of the C ABI gives what the CPU oracle gives: float32 sequence scores bit for bit, structure scores within 1e-6, the same hit positions."""
import io

import numpy as np
import pytest

from conftest import assert_f32_bits_equal, assert_struct_close
from test_gpu_library import _below_max, _clear_of
from test_gpu_parity import rand_stream, rand_struct_pssm, rand_table

pytestmark = pytest.mark.gpu

WIDTHS = [65, 100, 180, 181, 257]      # up to 180 a structure part runs k_profile itself, beyond it the plain k_wide


@pytest.mark.parametrize("m", WIDTHS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_wide_all_scores(ctx, oracle, m, dtype):
    rng = np.random.default_rng(7 * m)
    s = rand_stream(rng, 9, 0, 3 * m, dtype=dtype)              # records shorter than, as long as and longer than the motif
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m, inf_frac=0.05 if m == 100 else 0.0)
    motif = ctx.motif(T, P)
    ctx.stage(s.codes, s.profile)
    sq, st = ctx.scan_staged(motif)
    motif.close()
    assert_f32_bits_equal(sq, oracle.stream_seq(s.codes, T))
    assert_struct_close(st, oracle.stream_struct(s.profile, P))
    # each part alone
    motif = ctx.motif(T, None)
    ctx.stage(s.codes, None)
    sq2, none = ctx.scan_staged(motif)
    assert none is None
    assert_f32_bits_equal(sq2, sq)
    f64 = ctx.scan_letters_f64_host(motif, s.codes)
    motif.close()
    want = oracle.stream_letters_f64(s.codes, T)
    assert np.array_equal(np.isnan(f64), np.isnan(want)) and np.array_equal(f64[~np.isnan(want)], want[~np.isnan(want)])
    motif = ctx.motif(None, P)
    ctx.stage(None, s.profile)
    none, st2 = ctx.scan_staged(motif)
    motif.close()
    assert none is None and np.array_equal(np.isnan(st2), np.isnan(st)) and np.array_equal(st2[~np.isnan(st)], st[~np.isnan(st)])


@pytest.mark.parametrize("m", WIDTHS)
def test_wide_hits_every_mode(ctx, oracle, m):
    rng = np.random.default_rng(11 * m)
    s = rand_stream(rng, 12, m, 6 * m, foreign=0.0005)
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m)
    seq = oracle.stream_seq(s.codes, T)
    st = oracle.stream_struct(s.profile, P)
    fin_q = seq[np.isfinite(seq)].astype(np.float64)
    fin_t = st[np.isfinite(st)]
    thr_s = _clear_of(fin_q, _below_max(fin_q, 0.8))
    thr_t = _clear_of(fin_t, _below_max(fin_t, 0.6))
    # combined (one fused pass or letters first + verification: both go through the plain kernel for the letters)
    motif = ctx.motif(T, P)
    pos, hs, ht = ctx.hits_host(motif, s.codes, s.profile, thr_s, thr_t)
    want = oracle.stream_hits(seq, st, thr_s, thr_t)
    assert len(want) > 5 and np.array_equal(pos, want)
    assert_f32_bits_equal(hs, seq[want])
    assert_struct_close(ht, st[want])
    motif.close()
    # sequence only, structure only
    motif = ctx.motif(T, None)
    pos, hs, _ = ctx.hits_host(motif, s.codes, None, thr_s, -np.inf)
    want = oracle.stream_hits(seq, None, thr_s, -np.inf)
    assert len(want) > 5 and np.array_equal(pos, want)
    assert_f32_bits_equal(hs, seq[want])
    motif.close()
    motif = ctx.motif(None, P)
    pos, _, ht = ctx.hits_host(motif, None, s.profile, -np.inf, thr_t)
    want = oracle.stream_hits(None, st, -np.inf, thr_t)
    assert len(want) > 5 and np.array_equal(pos, want)
    assert_struct_close(ht, st[want])
    motif.close()


def test_wide_pfm_through_the_command_line(tmp_path):
    """a 70-wide sequence PFM through bin/rnascan's main(): the table equals the oracle engine's, byte for byte"""
    from engines import OracleEngine
    from rnascan_amd import cli, scanner
    rng = np.random.default_rng(70)
    pfm = tmp_path / "wide.pfm"
    with open(pfm, "w") as f:
        f.write("PO\tA\tC\tG\tU\n")
        for i, row in enumerate(rng.dirichlet(np.full(4, 0.6), size=70)):
            f.write("%d\t%s\n" % (i, "\t".join(repr(float(x)) for x in row)))
    fa = tmp_path / "s.fa"
    with open(fa, "w") as f:
        for i in range(6):
            f.write(">r%d\n%s\n" % (i, "".join(rng.choice(list("ACGU"), size=int(rng.integers(60, 400))))))
    argv = ["-p", str(pfm), "-C", "0.01", "-m", " -inf", str(fa)]
    want, got = io.StringIO(), io.StringIO()
    cli.main(argv, engine=OracleEngine(), out=want)
    eng = scanner.HipEngine(0)
    cli.main(argv, engine=eng, out=got)
    eng.close()
    assert want.getvalue().count("\n") > 100
    assert got.getvalue() == want.getvalue()


@pytest.mark.parametrize("m", [65, 128, 129, 640, 1500, 4096])
def test_wide_letters_slab_kernel_over_many_tiles(ctx, oracle, m, monkeypatch):
    """the slab kernel of letters-only wide PFMs (k_wide_letters: codes in LDS, the table through LDS 64 rows at a time) on a
    stream of many tiles, slab counts with and without a ragged last slab: float32 scores bit for bit, fp64 scores exactly
    (same order of additions), the same hits -- and the same as the plain one-thread-per-window kernel (PFMSCAN_WIDE_PLAIN)"""
    rng = np.random.default_rng(13 * m)
    s = rand_stream(rng, 7, m, 3 * m + 5000, foreign=0.0002)
    T = rand_table(rng, m)
    motif = ctx.motif(T, None)
    want32, want64 = oracle.stream_seq(s.codes, T), oracle.stream_letters_f64(s.codes, T)
    ctx.stage(s.codes, None)
    got32, _ = ctx.scan_staged(motif)
    assert_f32_bits_equal(got32, want32)
    got64 = ctx.scan_letters_f64_host(motif, s.codes)
    ok = ~np.isnan(want64)
    assert np.array_equal(np.isnan(got64), ~ok) and np.array_equal(got64[ok], want64[ok])
    fin = want32[np.isfinite(want32)].astype(np.float64)
    thr = _clear_of(fin, _below_max(fin, 0.7))
    pos, hs, _ = ctx.hits_host(motif, s.codes, None, thr, -np.inf)
    wpos = oracle.stream_hits(want32, None, thr, -np.inf)
    assert len(wpos) > 5 and np.array_equal(pos, wpos)
    assert_f32_bits_equal(hs, want32[wpos])
    pos64, sc64 = ctx.hits_letters_f64_host(motif, s.codes, thr)
    w64 = oracle.stream_hits(None, want64, -np.inf, thr)
    assert np.array_equal(pos64, w64) and np.array_equal(sc64, want64[w64])
    monkeypatch.setenv("PFMSCAN_WIDE_PLAIN", "1")
    ctx.stage(s.codes, None)
    plain32, _ = ctx.scan_staged(motif)
    assert_f32_bits_equal(plain32, got32)
    motif.close()
