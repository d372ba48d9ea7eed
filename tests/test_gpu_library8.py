"""LETTER libraries on the device (SURVEY 8f N1 x N4): every motif of a structure-letter library in one pass over an 8-code
stream (k_library8), and two-FASTA libraries -- sequence PFM k over the sequences AND structure-letter PFM k over the
structure strings of the same records -- in k_library with the second code stream as its structure side.

Reference semantics: matrix.py:25-43 (_py_calculate: fp64 sum, no float32 cast, NaN on an unknown letter), rnascan.py:263
(strict >), rnascan.py:416-434 (combine: both tables), pfmutil.py:89-133 (the multi-PFM format).  Checked against the
per-motif entry points (pfmscan_hits_letters_f64_host / pfmscan_hits_pair_host), which tests/test_gpu_letters8.py checks
against the CPU oracle, and against the oracle directly."""
import numpy as np
import pytest

from test_gpu_letters8 import _between, _stream, _table
from test_gpu_parity import rand_table

pytestmark = pytest.mark.gpu


def _per_motif(ctx, s, LT, thr):
    pos, mot, sc = [], [], []
    for k in range(LT.shape[0]):
        mo = ctx.motif(LT[k], None)
        p, v = ctx.hits_letters_f64_host(mo, s.codes, float(thr[k]))
        mo.close()
        pos.append(p)
        mot.append(np.full(p.size, k, dtype=np.int32))
        sc.append(v)
    pos, mot, sc = np.concatenate(pos), np.concatenate(mot), np.concatenate(sc)
    order = np.lexsort((mot, pos))
    return pos[order], mot[order], sc[order]


def _thresholds(oracle, s, LT, q):
    out = []
    for k in range(LT.shape[0]):
        full = oracle.stream_letters_f64(s.codes, LT[k])
        fin = full[np.isfinite(full)]
        out.append(_between(full, q) if np.unique(fin).size > 2 else 0.0)
    return np.array(out)


@pytest.mark.parametrize("n,m", [(1, 12), (2, 1), (7, 2), (8, 3), (9, 4), (16, 5), (17, 7), (24, 8), (40, 9), (19, 11), (128, 12), (129, 12),
                                 (300, 12), (13, 13), (11, 15), (20, 16), (10, 17), (9, 18), (8, 20), (7, 21), (6, 24), (5, 25), (9, 28),
                                 (33, 29), (12, 31), (70, 32)])
def test_letter_library_equals_per_motif_hits(ctx, oracle, n, m):
    """1 .. 300 motifs x widths 1 .. 32 (every padded row count of both buckets), lower-case codes, foreign letters"""
    rng = np.random.default_rng(7000 + 40 * n + m)
    s = _stream(rng, [5000, 0, m - 1, m, m + 1, 777, 4096, 9000, 3])
    LT = np.stack([_table(rng, m, neg_inf=0.03 if k % 3 == 0 else 0.0, nan=0.02 if k % 5 == 0 else 0.0) for k in range(n)])
    lib = ctx.library(None, struct_letters=LT)
    for q in (0.999, 0.97):
        thr = _thresholds(oracle, s, LT, q)
        pos, mot, sq, st = ctx.library_hits_letters_host(lib, s.codes, None, None, thr)
        wpos, wmot, wsc = _per_motif(ctx, s, LT, thr)
        assert sq is None
        assert np.array_equal(pos, wpos) and np.array_equal(mot, wmot), (n, m, q, pos.size, wpos.size)
        assert np.array_equal(st, wsc)                                  # the same sequential fp64 sum: bit-identical
    lib.close()


def test_letter_library_against_the_oracle_with_special_cells(ctx, oracle):
    """-inf / NaN cells get no credit, +inf cells switch a motif's prefilter off, thresholds ON scores (strict >),
    alphabets smaller than 7 letters"""
    rng = np.random.default_rng(11)
    m = 12
    s = _stream(rng, [30000, 2500, 11, 12, 13, 8000], foreign=0.01)
    LT = np.stack([_table(rng, m, neg_inf=0.1), _table(rng, m, nan=0.05), _table(rng, m, pos_inf=0.03),
                   _table(rng, m, neg_inf=0.05, nan=0.03, pos_inf=0.02), _table(rng, m), _table(rng, m, n_letters=4),
                   _table(rng, m, n_letters=2), _table(rng, m, scale=30.0), _table(rng, m, scale=1e-3)])
    lib = ctx.library(None, struct_letters=LT)
    full = [oracle.stream_letters_f64(s.codes, LT[k]) for k in range(LT.shape[0])]
    fin = [np.sort(f[np.isfinite(f)]) for f in full]
    for kind in ("on_score", "between", "low", "huge"):
        if kind == "on_score":
            thr = np.array([f[-5] if f.size > 5 else 0.0 for f in fin])
        elif kind == "between":
            thr = np.array([_between(f, 0.995) if np.unique(f).size > 2 else 0.0 for f in full])
        elif kind == "low":
            thr = np.full(LT.shape[0], -40.0)
        else:
            thr = np.full(LT.shape[0], 1e300)
        pos, mot, _, st = ctx.library_hits_letters_host(lib, s.codes, None, None, thr)
        for k in range(LT.shape[0]):
            sel = mot == k
            wpos = oracle.stream_hits(None, full[k], -np.inf, float(thr[k]))
            assert np.array_equal(pos[sel], wpos), (kind, k, int(sel.sum()), wpos.size)
            assert np.array_equal(st[sel], full[k][wpos])
            assert not np.isnan(st[sel]).any() and not np.isneginf(st[sel]).any()
    lib.close()


def test_letter_library_errors(ctx):
    rng = np.random.default_rng(3)
    LT = np.stack([_table(rng, 12) for _ in range(4)])
    lib = ctx.library(None, struct_letters=LT)
    s = _stream(rng, [2000])
    with pytest.raises(ValueError):                       # every window would be a hit
        ctx.library_hits_letters_host(lib, s.codes, None, None, -np.inf)
    with pytest.raises(ValueError):
        ctx.library_hits_letters_host(lib, s.codes, None, None, np.nan)
    lib.close()
    with pytest.raises(ValueError):                       # wider than the letter library kernel takes
        ctx.library(None, struct_letters=np.stack([_table(rng, 33)]))
    bad = LT.copy()
    bad[0, 0, 7] = 0.0                                    # the foreign column must be NaN
    with pytest.raises(ValueError):
        ctx.library(None, struct_letters=bad)
    empty = ctx.library(None, struct_letters=LT)
    pos, mot, _, st = ctx.library_hits_letters_host(empty, np.zeros(0, dtype=np.uint8), None, None, 1.0)
    assert pos.size == 0
    empty.close()


def _pair_stream(rng, lengths):
    from rnascan_amd import pack
    seqs, structs = [], []
    for L in lengths:
        c = rng.integers(0, 4, size=L).astype(np.uint8)
        c[rng.random(L) < 0.003] = pack.SEP
        t = rng.integers(0, 7, size=L).astype(np.uint8)
        t[rng.random(L) < 0.003] = pack.SEP
        low = (rng.random(L) < 0.3) & (t != pack.SEP)
        t[low] |= pack.CASE_BIT
        seqs.append(c)
        structs.append(t)
    a, b = pack.pack(seqs), pack.pack(structs)
    return a, b


@pytest.mark.parametrize("n,m", [(1, 12), (5, 1), (12, 7), (13, 8), (24, 12), (25, 16), (9, 17), (17, 18), (8, 31), (9, 32), (5, 33), (4, 64),
                                 (130, 12), (300, 9)])
def test_two_fasta_library_equals_per_pair_hits(ctx, oracle, n, m):
    """pair k = (sequence PFM k, structure-letter PFM k): float32-cast compare on the first stream, fp64 on the second"""
    rng = np.random.default_rng(9000 + 40 * n + m)
    a, b = _pair_stream(rng, [6000, 0, m - 1, m, m + 1, 1500, 8192, 3])
    LT = np.stack([rand_table(rng, m, inf_frac=0.05 if k % 4 == 0 else 0.0) for k in range(n)])
    ST = np.stack([_table(rng, m, neg_inf=0.03 if k % 3 == 0 else 0.0) for k in range(n)])
    lib = ctx.library(LT, struct_letters=ST)
    for q_seq, q_st in ((0.95, 0.6), (0.99, 0.2)):
        ts, tt = [], []
        for k in range(n):
            fs = oracle.stream_seq(a.codes, LT[k])
            ft = oracle.stream_letters_f64(b.codes, ST[k])
            ts.append(float(np.quantile(fs[np.isfinite(fs)], q_seq)) + 1e-4 if np.isfinite(fs).any() else 0.0)
            tt.append(_between(ft, q_st) if np.unique(ft[np.isfinite(ft)]).size > 2 else 0.0)
        ts, tt = np.array(ts), np.array(tt)
        pos, mot, sq, st = ctx.library_hits_letters_host(lib, a.codes, b.codes, ts, tt)
        wp, wm, wq, wt = [], [], [], []
        for k in range(n):
            ms, mt = ctx.motif(LT[k], None), ctx.motif(ST[k], None)
            p, vq, vt = ctx.hits_pair_host(ms, mt, a.codes, b.codes, float(ts[k]), float(tt[k]))
            ms.close()
            mt.close()
            # ... and the oracle's word on the same pair
            fs, ft = oracle.stream_seq(a.codes, LT[k]), oracle.stream_letters_f64(b.codes, ST[k])
            assert np.array_equal(p, oracle.stream_hits(fs, ft, float(ts[k]), float(tt[k])))
            wp.append(p)
            wm.append(np.full(p.size, k, dtype=np.int32))
            wq.append(vq)
            wt.append(vt)
        wp, wm, wq, wt = np.concatenate(wp), np.concatenate(wm), np.concatenate(wq), np.concatenate(wt)
        order = np.lexsort((wm, wp))
        assert np.array_equal(pos, wp[order]) and np.array_equal(mot, wm[order]), (n, m, pos.size, wp.size)
        assert np.array_equal(sq.view(np.uint32), wq[order].view(np.uint32))
        assert np.array_equal(st, wt[order])
    lib.close()


def test_letter_libraries_through_the_device_entry_point(ctx, oracle):
    """pfmscan_library_hits_letters_dev on torch tensors: unordered hits, total in *d_hit_count"""
    import ctypes
    import torch
    rng = np.random.default_rng(21)
    m, n = 12, 20
    a, b = _pair_stream(rng, [20000, 5000])
    LT = np.stack([rand_table(rng, m) for _ in range(n)])
    ST = np.stack([_table(rng, m) for _ in range(n)])
    dev = torch.device("cuda:0")
    d_a = torch.from_numpy(a.codes).to(dev)
    d_b = torch.from_numpy(b.codes).to(dev)
    cap = 1 << 16
    d_pos = torch.empty(cap, dtype=torch.int64, device=dev)
    d_mot = torch.empty(cap, dtype=torch.int32, device=dev)
    d_sq = torch.empty(cap, dtype=torch.float32, device=dev)
    d_st = torch.empty(cap, dtype=torch.float64, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    for kind in ("letters", "pair"):
        lib = ctx.library(LT if kind == "pair" else None, struct_letters=ST)
        ts = np.full(n, 3.0)
        tt = np.full(n, 2.0 if kind == "letters" else -3.0)
        first = d_a if kind == "pair" else d_b
        ctx._check(ctx._L.pfmscan_library_hits_letters_dev(
            ctx._h, lib._h, ctypes.c_void_p(first.data_ptr()), ctypes.c_void_p(d_b.data_ptr() if kind == "pair" else 0), a.codes.size,
            ts.ctypes.data_as(ctypes.c_void_p) if kind == "pair" else None, tt.ctypes.data_as(ctypes.c_void_p), cap,
            ctypes.c_void_p(d_pos.data_ptr()), ctypes.c_void_p(d_mot.data_ptr()), ctypes.c_void_p(d_sq.data_ptr()),
            ctypes.c_void_p(d_st.data_ptr()), ctypes.c_void_p(d_cnt.data_ptr()), None))
        ctx.synchronize()
        k = int(d_cnt.item())
        assert 0 < k <= cap
        got = sorted(zip(d_pos[:k].cpu().numpy().tolist(), d_mot[:k].cpu().numpy().tolist(), d_st[:k].cpu().numpy().tolist()))
        if kind == "pair":
            pos, mot, _, st = ctx.library_hits_letters_host(lib, a.codes, b.codes, ts, tt)
        else:
            pos, mot, _, st = ctx.library_hits_letters_host(lib, b.codes, None, None, tt)
        assert got == sorted(zip(pos.tolist(), mot.tolist(), st.tolist()))
        lib.close()


# ---- the CLI: `-q struct_library structs.fa` and `-p seq_library -q struct_library seqs.fa structs.fa` ---------------------
def _letter_libraries(tmp_path, n_pairs=21, seed=9):
    """two multi-PFM libraries (pfmutil.py:89-133) that share motif ids, mixed widths up to 32, + one structure PFM too wide
    for the library kernel (scanned on its own) and one without a partner; FASTA pair with the same records"""
    from test_gpu_letters8 import _fasta_pair
    from test_scanner_cpu import _write_multi_pfm
    rng = np.random.default_rng(seed)
    seq_m, st_m = [], []
    for k in range(n_pairs):
        w = int(rng.choice([5, 8, 8, 12, 12, 17, 32]))
        seq_m.append(("RBP%02d" % k, list("ACGU"), rng.dirichlet(np.full(4, 0.5), size=w)))
        st_m.append(("RBP%02d" % k, list("EHTBLRM"), rng.dirichlet(np.full(7, 0.5), size=w)))
    seq_m.append(("WIDE", list("ACGU"), rng.dirichlet(np.full(4, 0.5), size=40)))
    st_m.append(("WIDE", list("EHTBLRM"), rng.dirichlet(np.full(7, 0.5), size=40)))
    st_m.append(("STRUCT_ONLY", list("EHTBLRM"), rng.dirichlet(np.full(7, 0.5), size=8)))
    lib_s, lib_t = str(tmp_path / "seq_lib.pfm"), str(tmp_path / "struct_lib.pfm")
    _write_multi_pfm(lib_s, seq_m)
    _write_multi_pfm(lib_t, st_m)
    _fasta_pair(tmp_path / "s.fa", tmp_path / "t.fa", [int(x) for x in rng.integers(0, 1500, size=70)] + [4, 5, 31, 32, 33, 40], rng)
    return lib_s, lib_t, str(tmp_path / "s.fa"), str(tmp_path / "t.fa")


@pytest.fixture(scope="module")
def engine():
    from rnascan_amd import scanner
    e = scanner.HipEngine(0)
    yield e
    e.close()


def _run(argv, engine):
    import io
    from rnascan_amd import cli
    out = io.StringIO()
    cli.main(argv, engine=engine, out=out)
    return out.getvalue()


def _assert_same_table(got, want, what):
    """byte equality, reported as the first differing line (pytest's own diff of two multi-megabyte strings takes minutes)"""
    if got == want:
        return
    g, w = got.split("\n"), want.split("\n")
    for i in range(max(len(g), len(w))):
        a, b = (g[i] if i < len(g) else "<end>"), (w[i] if i < len(w) else "<end>")
        if a != b:
            raise AssertionError("%s: %d / %d lines, first difference at line %d:\n got  %s\n want %s" % (what, len(g), len(w), i, a, b))


def test_cli_structure_letter_library_equals_the_per_motif_run(engine, tmp_path, monkeypatch):
    """`rnascan -q struct_library structs.fa`: the one-pass library kernel prints the bytes of the per-motif run (the
    oracle-backed engine has no library entry point: its scan_records loops over the motifs, as the reference would)"""
    from engines import OracleEngine
    from rnascan_amd import scanner
    _, lib_t, _, t_fa = _letter_libraries(tmp_path)
    calls = []
    real = scanner.HipEngine.library_hits_letters
    monkeypatch.setattr(scanner.HipEngine, "library_hits_letters", lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
    for extra in (["-m", "1.5"], ["-m", "-2", "-C", "0.01"], []):
        argv = ["-q", lib_t, "-u"] + extra + [t_fa]
        got, want = _run(argv, engine), _run(argv, OracleEngine())
        _assert_same_table(got, want, extra)
    assert got.count("\n") > 20 and len(calls) >= 3 * 5          # one library launch per width (5, 8, 12, 17, 32), not per motif
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "600")          # several batches: same bytes
    _assert_same_table(_run(["-q", lib_t, "-u", "-m", "1.5", t_fa], engine), _run(["-q", lib_t, "-u", "-m", "1.5", t_fa], OracleEngine()), "batches")


def test_cli_two_fasta_libraries_equal_the_per_pair_run(engine, tmp_path, monkeypatch):
    """`rnascan -p seq_library -q struct_library seqs.fa structs.fa`: every pair of a width in ONE launch, same bytes as the
    per-pair run (scanner.pair_motifs: two libraries pair by motif id; the reference never holds more than one motif per
    side, rnascan.py:262)"""
    from engines import OracleEngine
    from rnascan_amd import scanner
    lib_s, lib_t, s_fa, t_fa = _letter_libraries(tmp_path)
    calls = []
    real = scanner.HipEngine.library_hits_letters
    monkeypatch.setattr(scanner.HipEngine, "library_hits_letters", lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1])
    for extra in (["-m", "-1"], ["-m", "0.5", "-C", "0.01"], []):
        argv = ["-p", lib_s, "-q", lib_t, "-u"] + extra + [s_fa, t_fa]
        got, want = _run(argv, engine), _run(argv, OracleEngine())
        _assert_same_table(got, want, extra)
        if extra == ["-m", "-1"]:
            assert got.count("\n") > 20
    assert len(calls) >= 3 * 5


def test_letter_library_info_reports_its_passes(ctx):
    """128 structure-letter PFMs of width 12 fit the LDS at once (16 groups of 8), 300 need three passes"""
    rng = np.random.default_rng(2)
    for n, passes in ((128, 1), (129, 2), (300, 3)):
        lib = ctx.library(None, struct_letters=np.stack([_table(rng, 12) for _ in range(n)]))
        info = lib.info()
        assert info["n_motifs"] == n and info["m"] == 12 and info["passes"] == passes and info["motifs_per_pass"] == 128
        lib.close()
