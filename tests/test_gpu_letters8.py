"""SURVEY 8f N4 on the device: thresholded hits of generic-alphabet letter scans (fp64 compare and score,
matrix.py:25-43 + rnascan.py:263) and the two-FASTA combined scan (rnascan.py:119-123, :416-434), through the C ABI,
against the CPU oracle, the reference's `_py_calculate` goldens and the oracle-backed engine of the CLI."""
import io

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STRUCT = "EHTBLRM"


def _table(rng, m, n_letters=7, neg_inf=0.0, nan=0.0, pos_inf=0.0, scale=2.0):
    T = np.full((m, 8), np.nan)
    T[:, :n_letters] = rng.normal(-0.5, scale, size=(m, n_letters))
    r = rng.random((m, n_letters))
    T[:, :n_letters][r < neg_inf] = -np.inf
    T[:, :n_letters][(r >= neg_inf) & (r < neg_inf + nan)] = np.nan
    T[:, :n_letters][(r >= neg_inf + nan) & (r < neg_inf + nan + pos_inf)] = np.inf
    return T


def _stream(rng, lengths, n_letters=7, foreign=0.002, case=True):
    from rnascan_amd import pack
    codes = []
    for L in lengths:
        c = rng.integers(0, n_letters, size=L).astype(np.uint8)
        c[rng.random(L) < foreign] = pack.SEP
        if case:                                           # the case bit must not change any score
            low = (rng.random(L) < 0.3) & (c != pack.SEP)
            c[low] |= pack.CASE_BIT
        codes.append(c)
    return pack.pack(codes)


def _want(oracle, stream, T, thr):
    sc = oracle.stream_letters_f64(stream.codes, T)
    pos = oracle.stream_hits(None, sc, -np.inf, thr)
    return pos, sc[pos]


def _between(scores, q):
    """a threshold between two neighbouring scores near quantile q (never ON a score here)"""
    s = np.unique(scores[np.isfinite(scores)])
    k = min(max(int(q * s.size), 0), s.size - 2)
    return 0.5 * (s[k] + s[k + 1])


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 16, 17, 18, 23, 24, 25, 31, 32, 33, 40, 64, 65, 100])
def test_hits_letters_f64_every_width(ctx, oracle, m):
    rng = np.random.default_rng(100 + m)
    s = _stream(rng, [5000, 0, m - 1, m, m + 1, 777, 4096, 12000, 3])
    T = _table(rng, m, neg_inf=0.03 if m % 3 == 0 else 0.0)
    mo = ctx.motif(T, None)
    full = oracle.stream_letters_f64(s.codes, T)
    fin = full[np.isfinite(full)]
    thrs = [-np.inf, np.inf, _between(full, 0.5), _between(full, 0.999)]
    if fin.size:
        thrs += [float(np.sort(fin)[-3]), float(fin.max())]              # ON a score: strict `>` keeps only the larger ones
    for thr in thrs:
        pos, sc = ctx.hits_letters_f64_host(mo, s.codes, thr)
        wpos, wsc = _want(oracle, s, T, thr)
        assert np.array_equal(pos, wpos), (m, thr, pos.size, wpos.size)
        assert np.array_equal(sc, wsc), (m, thr)                        # same sequential fp64 sum: bit-identical
    mo.close()


@pytest.mark.parametrize("nj", [4, 8, 10, 12, 16])
def test_hits_letters_f64_every_bucket_of_the_prefilter(ctx, oracle, nj, monkeypatch):
    """every instantiation of k_letters_cred8 on PFMs narrower than it is sized for (rows beyond the width carry no
    credit).  The widest one once came out of the compiler wrong: its position loop was left rolled, pk[] became a register
    tuple indexed at run time, and the if-converted guarded update wrote outside it (profiles/r5/NOTES.md).  The loops are
    template-expanded since (static_for), the kernel is built WITHOUT a register cap again, and this is the test that showed it."""
    monkeypatch.setenv("PFMSCAN_CRED8_NJ", str(nj))
    for m in (3, 2 * nj - 1, 2 * nj):
        rng = np.random.default_rng(1000 + 10 * nj + m)
        s = _stream(rng, [9000, m, 5000, 31])
        T = _table(rng, m)
        full = oracle.stream_letters_f64(s.codes, T)
        mo = ctx.motif(T, None)
        for q in (0.999, 0.98):
            thr = _between(full, q)
            pos, sc = ctx.hits_letters_f64_host(mo, s.codes, thr)
            wpos, wsc = _want(oracle, s, T, thr)
            assert np.array_equal(pos, wpos), (nj, m, q, pos.size, wpos.size)
            assert np.array_equal(sc, wsc)
        mo.close()


def test_hits_letters_f64_special_cells_and_dense_thresholds(ctx, oracle):
    """-inf / NaN cells get no credit, +inf cells switch the prefilter off, dense thresholds take the exact kernel"""
    rng = np.random.default_rng(7)
    s = _stream(rng, [30000, 2500, 11, 12, 13, 8000], foreign=0.01)
    for kind in ("neg_inf", "nan", "pos_inf", "mixed"):
        T = _table(rng, 12, neg_inf=0.1 if kind in ("neg_inf", "mixed") else 0, nan=0.05 if kind in ("nan", "mixed") else 0,
                   pos_inf=0.03 if kind in ("pos_inf", "mixed") else 0)
        mo = ctx.motif(T, None)
        full = oracle.stream_letters_f64(s.codes, T)
        for thr in (-np.inf, -1e300, -40.0, -5.0, 0.0, 4.0, 1e300):
            pos, sc = ctx.hits_letters_f64_host(mo, s.codes, thr)
            wpos, wsc = _want(oracle, s, T, thr)
            assert np.array_equal(pos, wpos), (kind, thr, pos.size, wpos.size)
            assert np.array_equal(sc, wsc)
            assert not np.isnan(sc).any() and not np.isneginf(sc).any()       # NaN and -inf never pass, not even at -inf
        mo.close()
        del full


def test_hits_letters_f64_alphabets_of_other_sizes(ctx, oracle):
    """any alphabet of up to 7 letters (the unused columns are NaN), 4 letters included: fp64 semantics, no float32 cast"""
    rng = np.random.default_rng(9)
    for n_letters in (1, 2, 4, 5, 7):
        s = _stream(rng, [6000, 100, 9], n_letters=n_letters)
        T = _table(rng, 10, n_letters=n_letters)
        mo = ctx.motif(T, None)
        full = oracle.stream_letters_f64(s.codes, T)
        thr = _between(full, 0.98)
        pos, sc = ctx.hits_letters_f64_host(mo, s.codes, thr)
        wpos, wsc = _want(oracle, s, T, thr)
        assert np.array_equal(pos, wpos) and np.array_equal(sc, wsc)
        mo.close()


def test_hits_letters_f64_thresholds_a_float32_cast_would_move(ctx, oracle):
    """a window whose fp64 score lies just above the threshold while its float32 cast lies below (and the reverse) is
    decided in fp64, as Python floats are (matrix.py:25-43)"""
    rng = np.random.default_rng(21)
    s = _stream(rng, [20000])
    T = _table(rng, 12)
    full = oracle.stream_letters_f64(s.codes, T)
    fin = np.sort(full[np.isfinite(full)])
    x = fin[int(0.999 * fin.size)]
    mo = ctx.motif(T, None)
    for thr in (np.nextafter(x, -np.inf), x, np.nextafter(x, np.inf), float(np.float32(x))):
        pos, sc = ctx.hits_letters_f64_host(mo, s.codes, thr)
        wpos, wsc = _want(oracle, s, T, thr)
        assert np.array_equal(pos, wpos) and np.array_equal(sc, wsc)
    mo.close()


def test_hits_letters_f64_multi_tile_walk_and_queue_flushes(ctx, oracle):
    """a stream long enough for several tiles per workgroup and a threshold loose enough that the wave queues flush"""
    rng = np.random.default_rng(5)
    s = _stream(rng, [3000] * 400)
    T = _table(rng, 12, scale=1.0)
    full = oracle.stream_letters_f64(s.codes, T)
    mo = ctx.motif(T, None)
    for q in (0.9999, 0.99, 0.975, 0.9):                    # the last ones are dense: the exact kernel
        thr = _between(full, q)
        pos, sc = ctx.hits_letters_f64_host(mo, s.codes, thr)
        wpos, wsc = _want(oracle, s, T, thr)
        assert np.array_equal(pos, wpos), (q, pos.size, wpos.size)
        assert np.array_equal(sc, wsc)
    mo.close()


def test_hits_letters_f64_capacity_protocol_and_empty(ctx, oracle):
    from rnascan_amd import _lib
    rng = np.random.default_rng(2)
    s = _stream(rng, [4000, 4000])
    T = _table(rng, 8)
    mo = ctx.motif(T, None)
    wpos, _ = _want(oracle, s, T, -3.0)
    with pytest.raises(_lib.CapacityError) as e:
        ctx.hits_letters_f64_host(mo, s.codes, -3.0, capacity=10)
    assert e.value.required >= wpos.size
    pos, sc = ctx.hits_letters_f64_host(mo, s.codes, -3.0, capacity=int(e.value.required))
    assert np.array_equal(pos, wpos)
    pos, sc = ctx.hits_letters_f64_host(mo, np.zeros(0, dtype=np.uint8), 0.0)
    assert pos.size == 0 and sc.size == 0
    with pytest.raises(ValueError):
        ctx.hits_letters_f64_host(mo, s.codes, float("nan"))
    mo.close()


def test_py_calculate_goldens_through_the_hits_path(ctx, golden):
    """the reference's own `_py_calculate` outputs (tests/golden/make_golden.py ran the function unmodified): the windows
    the device reports above a threshold, and their scores, are the golden ones"""
    from rnascan_amd import pack
    for case in golden["py_calculate"]:
        letters = case["letters"]
        m = case["m"]
        T = np.full((len(case["table"]), 8), np.nan)
        T[:, :len(letters)] = np.array(case["table"], dtype=np.float64)
        want = np.array(case["scores"], dtype=np.float64)
        codes = np.append(pack.encode_letters(case["sequence"], letters, keep_case=True), pack.SEP).astype(np.uint8)
        mo = ctx.motif(T[:m], None)
        fin = np.sort(want[np.isfinite(want)])
        thrs = [-np.inf] + ([0.5 * (fin[fin.size // 2] + fin[fin.size // 2 - 1])] if fin.size > 1 else [])
        for thr in thrs:
            pos, sc = ctx.hits_letters_f64_host(mo, codes, thr)
            keep = np.flatnonzero(want > thr)
            assert np.array_equal(pos, keep), case["name"]
            assert np.array_equal(sc, want[keep]), case["name"]
        mo.close()


@pytest.mark.parametrize("m", [1, 6, 12, 18, 32, 40, 70])
def test_hits_pair_vs_oracle(ctx, oracle, m):
    from rnascan_amd import pack
    rng = np.random.default_rng(300 + m)
    lengths = [4000, 0, m, m - 1, 9000, 600]
    s1 = _stream(rng, lengths, n_letters=4, case=False, foreign=0.003)
    s2 = _stream(rng, lengths, n_letters=7, foreign=0.003)
    assert np.array_equal(s1.offsets, s2.offsets)
    T1, T2 = _table(rng, m, n_letters=4, scale=1.5), _table(rng, m, scale=1.5, neg_inf=0.02)
    a, b = ctx.motif(T1, None), ctx.motif(T2, None)
    sq = oracle.stream_seq(s1.codes, T1)
    st = oracle.stream_letters_f64(s2.codes, T2)
    for t1, t2 in ((_between(sq.astype(np.float64), 0.9), _between(st, 0.5)), (_between(sq.astype(np.float64), 0.3), _between(st, 0.3)),
                   (-np.inf, _between(st, 0.99)), (_between(sq.astype(np.float64), 0.99), -np.inf), (np.inf, 0.0)):
        pos, gq, gt = ctx.hits_pair_host(a, b, s1.codes, s2.codes, t1, t2)
        wpos = oracle.stream_hits(sq, st, t1, t2)
        assert np.array_equal(pos, wpos), (m, t1, t2, pos.size, wpos.size)
        assert np.array_equal(gq.view(np.uint32), sq[wpos].view(np.uint32)) and np.array_equal(gt, st[wpos])
    with pytest.raises(ValueError):
        ctx.hits_pair_host(a, ctx.motif(_table(rng, m + 1), None), s1.codes, s2.codes, 0.0, 0.0)
    a.close()
    b.close()


def test_dev_pointer_forms(ctx, oracle):
    """pfmscan_hits_letters_f64_dev / pfmscan_hits_pair_dev on torch-owned device buffers"""
    import torch
    from rnascan_amd import _lib
    rng = np.random.default_rng(4)
    lengths = [7000, 500, 12]
    s1 = _stream(rng, lengths, n_letters=4, case=False)
    s2 = _stream(rng, lengths)
    T1, T2 = _table(rng, 12, n_letters=4), _table(rng, 12)
    a, b = ctx.motif(T1, None), ctx.motif(T2, None)
    dev = torch.device("cuda:%d" % ctx.device)
    d1, d2 = torch.from_numpy(s1.codes).to(dev), torch.from_numpy(s2.codes).to(dev)
    cap = 4096
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hq = torch.empty(cap, dtype=torch.float32, device=dev)
    ht = torch.empty(cap, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    st = oracle.stream_letters_f64(s2.codes, T2)
    thr = _between(st, 0.99)
    ctx.hits_letters_f64_dev(b, d2.data_ptr(), s2.n_pos, thr, cap, hp.data_ptr(), ht.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    wpos = oracle.stream_hits(None, st, -np.inf, thr)
    order = torch.argsort(hp[:k]).cpu().numpy()
    assert k == wpos.size and np.array_equal(hp[:k].cpu().numpy()[order], wpos) and np.array_equal(ht[:k].cpu().numpy()[order], st[wpos])
    sq = oracle.stream_seq(s1.codes, T1)
    cnt.zero_()
    torch.cuda.synchronize()
    ctx.hits_pair_dev(a, b, d1.data_ptr(), d2.data_ptr(), s1.n_pos, 0.0, _between(st, 0.5), cap, hp.data_ptr(), hq.data_ptr(),
                      ht.data_ptr(), cnt.data_ptr())
    ctx.synchronize()
    k = int(cnt.item())
    wpos = oracle.stream_hits(sq, st, 0.0, _between(st, 0.5))
    order = torch.argsort(hp[:k]).cpu().numpy()
    assert k == wpos.size and np.array_equal(hp[:k].cpu().numpy()[order], wpos)
    assert np.array_equal(hq[:k].cpu().numpy()[order], sq[wpos]) and np.array_equal(ht[:k].cpu().numpy()[order], st[wpos])
    a.close()
    b.close()


# ---- the CLI in SS mode and in two-FASTA RNASS mode on the HIP engine ---------------------------------------------
def _pfm(path, letters, m, rng):
    with open(path, "w") as f:
        f.write("PO\t" + "\t".join(letters) + "\n")
        for j in range(m):
            f.write(str(j) + "\t" + "\t".join("%.5f" % x for x in rng.dirichlet(np.full(len(letters), 0.4))) + "\n")


def _fasta_pair(seq_path, struct_path, lengths, rng, crlf_every=4, foreign_every=5):
    """the same record ids in both files; the structure strings mix cases (reported as written) and every few records
    hold a foreign letter"""
    with open(seq_path, "wb") as f, open(struct_path, "wb") as g:
        for i, L in enumerate(lengths):
            s = "".join(rng.choice(list("ACGTUacgu"), size=L))
            t = "".join(rng.choice(list("EHTBLRMehtblrm"), size=L))
            if foreign_every and i % foreign_every == 0 and L > 30:
                s = s[:17] + "N" + s[18:]
                t = t[:25] + "x" + t[26:]
            eol = "\r\n" if crlf_every and i % crlf_every == 0 else "\n"
            f.write((">rec%d seq %d" % (i, i) + eol + eol.join(s[k:k + 70] for k in range(0, max(L, 1), 70)) + eol).encode())
            g.write((">rec%d struct\t\"%d\"" % (i, i) + eol + eol.join(t[k:k + 50] for k in range(0, max(L, 1), 50)) + eol).encode())


@pytest.fixture(scope="module")
def engine():
    from rnascan_amd import scanner
    e = scanner.HipEngine(0)
    yield e
    e.close()


def _run(argv, engine):
    from rnascan_amd import cli
    out = io.StringIO()
    cli.main(argv, engine=engine, out=out)
    return out.getvalue()


def test_cli_ss_mode_on_the_gpu_equals_the_oracle_engine_run(engine, tmp_path):
    """`rnascan -q pfm structs.fa` (README usage, rnascan.py:124-133): same bytes from the HIP engine and from the
    oracle-backed engine, at the default threshold, a loose one and -m ' -inf'"""
    from engines import OracleEngine
    rng = np.random.default_rng(31)
    _pfm(tmp_path / "st.pfm", STRUCT, 9, rng)
    _fasta_pair(tmp_path / "s.fa", tmp_path / "t.fa", [int(x) for x in rng.integers(0, 900, size=60)] + [8, 9, 10], rng)
    for extra in ([], ["-m", "1.5"], ["-m", " -inf"], ["-m", "-4", "-C", "0.01"]):
        argv = ["-q", str(tmp_path / "st.pfm"), "-u"] + extra + [str(tmp_path / "t.fa")]
        got, want = _run(argv, engine), _run(argv, OracleEngine())
        assert got == want, extra
        assert got.count("\n") > 1 or extra == []
    # background estimated from the structure FASTA itself (no -u): counted natively, same table
    argv = ["-q", str(tmp_path / "st.pfm"), "-m", "1", "-C", "0.5", str(tmp_path / "t.fa")]
    assert _run(argv, engine) == _run(argv, OracleEngine())


def test_cli_two_fasta_rnass_on_the_gpu_is_combine_of_the_two_tables(engine, tmp_path, monkeypatch):
    """`rnascan -p pfm -q pfm seqs.fa structs.fa`: the fused two-stream scan prints the bytes of combine() of the two
    single tables (rnascan.py:416-434) -- made here by the oracle-backed engine through the reference-shaped path"""
    from engines import OracleEngine
    from rnascan_amd import cli
    rng = np.random.default_rng(32)
    _pfm(tmp_path / "sq.pfm", "ACGU", 7, rng)
    _pfm(tmp_path / "st.pfm", STRUCT, 7, rng)
    _fasta_pair(tmp_path / "s.fa", tmp_path / "t.fa", [int(x) for x in rng.integers(0, 1200, size=80)] + [6, 7, 8], rng)
    for extra in (["-m", "-1"], ["-m", " -inf"], ["-m", "0.5", "-C", "0.01"], []):
        argv = ["-p", str(tmp_path / "sq.pfm"), "-q", str(tmp_path / "st.pfm"), "-u"] + extra + [str(tmp_path / "s.fa"), str(tmp_path / "t.fa")]
        got = _run(argv, engine)
        with monkeypatch.context() as mp:
            mp.setattr(cli, "_same_records", lambda *a: None)        # the reference's shape: two tables + combine()
            want = _run(argv, OracleEngine())
        assert got == want, extra
    assert got.startswith("Sequence_ID\tDescription.Seq\tMotif_ID.Seq\tStart\tEnd\tSequence.Seq\tLogOdds.Seq\tDescription.Struct")


def test_cli_two_fasta_small_batches_and_a_length_mismatch(engine, tmp_path, monkeypatch):
    """several batches per run, and one record whose two strings differ in length: that batch falls back to the two
    tables + join, the others stay fused -- same bytes either way"""
    from engines import OracleEngine
    from rnascan_amd import cli
    rng = np.random.default_rng(33)
    _pfm(tmp_path / "sq.pfm", "ACGU", 6, rng)
    _pfm(tmp_path / "st.pfm", STRUCT, 6, rng)
    _fasta_pair(tmp_path / "s.fa", tmp_path / "t.fa", [300] * 30, rng, foreign_every=0)
    with open(tmp_path / "t.fa", "ab") as g, open(tmp_path / "s.fa", "ab") as f:
        f.write(b">odd one\n" + b"ACGU" * 30 + b"\n")
        g.write(b">odd one\n" + b"EHTL" * 20 + b"\n")
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "256")             # 8 x 256 positions per launch: ~7 records a batch
    argv = ["-p", str(tmp_path / "sq.pfm"), "-q", str(tmp_path / "st.pfm"), "-u", "-m", "-2", str(tmp_path / "s.fa"), str(tmp_path / "t.fa")]
    got = _run(argv, engine)
    with monkeypatch.context() as mp:
        mp.setattr(cli, "_same_records", lambda *a: None)
        want = _run(argv, OracleEngine())
    assert got == want and got.count("\n") > 10
