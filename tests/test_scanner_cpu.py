"""Table layer and CLI on a GPU-less host.  Scores come from the TEST-ONLY
OracleEngine (tests/engines.py); what is checked here is the host logic: record
packing, thresholds, 1-based coordinates, rounding, column order, the fused
combined scan against the reference's own combine() output."""
import io
import os
import shutil

import numpy as np
import pandas as pd
import pytest

from conftest import DATA_DIR, nasty_fasta
from engines import OracleEngine
from rnascan_amd import cli, fasta, pack, pssm, scanner

SEQ_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt")
STRUCT_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt")
HIST_FA = os.path.join(DATA_DIR, "HIST2H3C_3p_end.fa")
HIST_PROFILE = os.path.join(DATA_DIR, "HIST2H3C_3p_end_structure.txt")


@pytest.fixture()
def avgdir(tmp_path):
    d = tmp_path / "avg"
    d.mkdir()
    shutil.copyfile(HIST_PROFILE, d / "structure.hg19_dna.txt")
    return str(d)


def hist_bg():
    rec = list(fasta.parse_sequences(HIST_FA))[0]
    return {l: c for l, c in fasta.compute_background(HIST_FA, fasta.RNA, verbose=False).items()}, rec


def test_scan_records_matches_reference_seq_table(golden):
    bg, rec = hist_bg()
    P = {"SLBP_seq": pssm.pfm2pssm(SEQ_PFM, 0.01, fasta.RNA, bg)}
    df = scanner.scan_records(OracleEngine(), [rec], P, fasta.RNA, 0.0)
    want = golden["combine"]["seq_rows"]
    assert list(df.columns) == ["Sequence_ID", "Description", "Motif_ID", "Start", "End", "Sequence", "LogOdds"]
    assert df["LogOdds"].dtype == np.float32 == np.dtype(golden["combine"]["seq_logodds_dtype"])
    got = [[r[0], r[1], r[2], int(r[3]), int(r[4]), r[5], float(r[6])] for r in df.itertuples(index=False)]
    assert got == want


def test_scan_profile_dir_matches_reference_scan_main(golden, avgdir):
    P = {"SLBP_struct": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
    df = scanner.scan_profile_dir(OracleEngine(), avgdir, P, 0.0, "aligned", np.float64)
    g = golden["scan_main_dir"]
    assert list(df.columns) == g["columns"]
    got = [[r[0], r[1], r[2], int(r[3]), int(r[4]), r[5], float(r[6])] for r in df.itertuples(index=False)]
    assert len(got) == len(g["rows"])
    for a, b in zip(got, g["rows"]):
        assert a[:6] == b[:6] and abs(a[6] - b[6]) <= 1e-9


def assert_tsv_equal(got, want, float_cols=("LogOdds.Struct", "LogOdds.SeqStruct"), tol=1e-9):
    """every field identical as TEXT except the unrounded fp64 structure columns, which the
    reference computes through BLAS ddot (summation order unspecified): those within tol"""
    g, w = [l.split("\t") for l in got.splitlines()], [l.split("\t") for l in want.splitlines()]
    assert g[0] == w[0]
    assert len(g) == len(w)
    fc = [g[0].index(c) for c in float_cols if c in g[0]]
    for a, b in zip(g[1:], w[1:]):
        assert len(a) == len(b)
        for k, (x, y) in enumerate(zip(a, b)):
            if k in fc:
                assert abs(float(x) - float(y)) <= tol, (x, y)
            else:
                assert x == y, (g[0][k], x, y)


def test_combined_tsv_matches_reference_output(golden, avgdir):
    """rnascan -p SEQ -q STRUCT -C 0.01 -m 0 -B <uniform> fasta avgdir: the TSV the
    reference's scan_main + combine + _add_match_id + to_csv produced"""
    out = io.StringIO()
    argv = ["-p", SEQ_PFM, "-q", STRUCT_PFM, "-C", "0.01", "-m", "0", HIST_FA, avgdir]
    # sequence background computed from the FASTA (as make_golden did); structure side
    # needs a uniform background file because -u would also make the sequence side uniform
    bgfile = os.path.join(os.path.dirname(avgdir), "bg_struct.txt")
    open(bgfile, "w").write(repr({l: 1.0 / 7 for l in fasta.STRUCT}))
    os.symlink(SEQ_PFM, os.path.join(os.path.dirname(avgdir), "SLBP_seq.txt"))
    os.symlink(STRUCT_PFM, os.path.join(os.path.dirname(avgdir), "SLBP_struct.txt"))
    argv[1] = os.path.join(os.path.dirname(avgdir), "SLBP_seq.txt")
    argv[3] = os.path.join(os.path.dirname(avgdir), "SLBP_struct.txt")
    cli.main(argv + ["-B", bgfile], engine=OracleEngine(), out=out)
    assert_tsv_equal(out.getvalue(), golden["combine"]["tsv"])


def test_fused_combined_equals_two_tables_plus_join(avgdir):
    bg, rec = hist_bg()
    eng = OracleEngine()
    sp = {"s": pssm.pfm2pssm(SEQ_PFM, 0.01, fasta.RNA, bg)}
    tp = {"t": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
    named = scanner.load_profile_dir(avgdir)
    for thr in (-np.inf, -5.0, 0.0, 6.0):
        fused = scanner.scan_combined(eng, [rec], named, sp, tp, thr, "aligned", np.float64)
        a = scanner.scan_records(eng, [rec], sp, fasta.RNA, thr)
        b = scanner.scan_profiles(eng, named, tp, thr, "aligned", np.float64)
        joined = scanner.combine(a, b)
        assert list(fused.columns) == list(joined.columns)
        pd.testing.assert_frame_equal(fused.reset_index(drop=True), joined.reset_index(drop=True), check_dtype=False)


def test_fused_combined_declines_unpairable_inputs(avgdir):
    bg, rec = hist_bg()
    sp = {"s": pssm.pfm2pssm(SEQ_PFM, 0.01, fasta.RNA, bg)}
    tp = {"t": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
    named = scanner.load_profile_dir(avgdir)
    short = fasta.Record(rec.id, rec.description, rec.seq[:100])
    assert scanner.scan_combined(OracleEngine(), [short], named, sp, tp, 0.0) is None
    assert scanner.scan_combined(OracleEngine(), [rec, rec], named, sp, tp, 0.0) is None
    other = fasta.Record("nobody", "nobody", rec.seq)
    assert len(scanner.scan_combined(OracleEngine(), [other], named, sp, tp, 0.0)) == 0


def test_pairing_switch_reproduces_both_reference_behaviours(golden):
    cases = {c["name"]: c for c in golden["scan_averaged_structure"]}
    P = {"m": pssm.pfm2pssm(STRUCT_PFM, 0.0, fasta.STRUCT, None)}
    for pairing, name in (("positional", "hist_slbp_pc0_positional"), ("aligned", "hist_slbp_pc0_aligned")):
        df = scanner.scan_averaged_structure(OracleEngine(), HIST_PROFILE, P, -np.inf, pairing)
        want = cases[name]["rows"]
        assert df["Start"].tolist() == [r[0] for r in want] and df["End"].tolist() == [r[1] for r in want]
        assert np.allclose(df["LogOdds"].to_numpy(), [r[2] for r in want], rtol=0, atol=1e-9)
        assert list(df.columns) == cases[name]["columns"] and (df["Sequence"] == ".").all()


def test_minus_inf_threshold_drops_nan_and_minus_inf_windows():
    rec = fasta.Record("r", "r", "ACGUNACGUACGUACGU")
    T = {"m": pssm.PSSM("GAUC", {"A": [0.5, -np.inf], "C": [0.1, 0.2], "G": [0.3, 0.1], "U": [-0.2, 0.4]})}
    df = scanner.scan_records(OracleEngine(), [rec], T, fasta.RNA, float("-inf"))
    starts = df["Start"].tolist()
    assert 4 not in starts and 5 not in starts          # windows covering N
    for s in starts:                                     # second letter A gives -inf: dropped
        assert rec.seq[s] != "A"
    assert len(starts) > 0


def test_struct_letter_string_mode_rounds_like_python():
    P = {"m": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
    rec = fasta.Record("s", "s", "EEEEEEEELLLLHHHHHHLLLLRRRREEEEEEEEEEEEE")
    df = scanner.scan_records(OracleEngine(), [rec], P, fasta.STRUCT, -1000.0)
    assert df["LogOdds"].dtype == np.float64 and len(df) == len(rec.seq) - 18 + 1
    assert all(round(x, 3) == x for x in df["LogOdds"])
    assert df["Sequence"].iloc[0] == rec.seq[:18]


def test_cli_modes_and_errors(tmp_path, capsys):
    with pytest.raises(SystemExit):
        cli.main([HIST_FA], engine=OracleEngine())                      # no PFM
    with pytest.raises(SystemExit):
        cli.main(["-p", SEQ_PFM, "-u", "-b", "x", HIST_FA], engine=OracleEngine())
    with pytest.raises(SystemExit):
        cli.main(["-p", SEQ_PFM, "-q", STRUCT_PFM, HIST_FA], engine=OracleEngine())   # two PFMs, one file
    out = io.StringIO()
    cli.main(["-p", SEQ_PFM, "-u", "-m", "8", HIST_FA], engine=OracleEngine(), out=out)
    lines = out.getvalue().splitlines()
    assert lines[0].split("\t") == ["Sequence_ID", "Description", "Motif_ID", "Start", "End", "Sequence", "LogOdds", "Match_ID"]
    assert len(lines) == 2 and lines[1].split("\t")[3:8] == ["213", "230", "AAAGGCUCUUUUCAGAGC", "14.259", "1"]
    out = io.StringIO()
    cli.main(["-p", SEQ_PFM, "-t", "AAAGGCTCTTTTCAGAGCaa", "-m", "3"], engine=OracleEngine(), out=out)
    row = out.getvalue().splitlines()[1].split("\t")
    assert row[0] == "testseq" and row[3:7] == ["1", "18", "AAAGGCUCUUUUCAGAGC", "14.259"]
    with pytest.raises(SystemExit):
        cli.main(["-p", SEQ_PFM, "-g", HIST_FA], engine=OracleEngine(), out=io.StringIO())   # --bgonly prints and exits
    err = capsys.readouterr().err
    assert "Loading PFM" in err and "Found 1 motifs" in err and "Scanning sequences" in err and "Processed 1 sequences" in err


def test_profile_store_roundtrip_and_scan(tmp_path, avgdir):
    """N2: packed profile store == parsing the text files, through scanner and CLI"""
    from rnascan_amd import store
    rng = np.random.default_rng(5)
    d = tmp_path / "many"
    d.mkdir()
    shutil.copyfile(HIST_PROFILE, d / "structure.hg19_dna.txt")
    for i in range(6):
        L = int(rng.integers(5, 80))
        prof = rng.dirichlet(np.full(7, 0.3), size=L)
        with open(d / ("structure.r%d.txt" % i), "w") as f:
            f.write("PO\t" + "\t".join("BEHLMRT") + "\n")
            for k, row in enumerate(prof):
                f.write(str(k) + "\t" + "\t".join(str(float(x)) for x in row) + "\n")
    sdir = str(tmp_path / "store")
    assert store.main([str(d), sdir]) == 0 and store.is_store(sdir) and not store.is_store(str(d))
    ps = store.ProfileStore(sdir)
    assert ps.ids == sorted(ps.ids) and len(ps.ids) == 7 and ps.letters == list("BEHLMRT")
    letters, prof = fasta.read_profile(HIST_PROFILE)
    k = ps.ids.index("hg19_dna")
    assert np.array_equal(np.asarray(ps.named()[k][2]), prof)
    st = ps.stream()
    assert st.n_pos == ps.n_pos and np.array_equal(st.profile[ps.offsets + ps.lengths], np.zeros((7, 7)))
    sub = ps.stream(2, 5)
    assert sub.n_records == 3 and sub.offsets[0] == 0 and sub.n_pos == int((ps.lengths[2:5] + 1).sum())
    P = {"SLBP_struct": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
    a = scanner.scan_store(OracleEngine(), ps, P, -10.0)
    named = sorted(scanner.load_profile_dir(str(d)))
    b = scanner.scan_profiles(OracleEngine(), named, P, -10.0, "aligned", np.float64)
    pd.testing.assert_frame_equal(a, b)
    o1, o2 = io.StringIO(), io.StringIO()
    cli.main(["-q", STRUCT_PFM, "-u", "-m", "-10", sdir], engine=OracleEngine(), out=o1)
    cli.main(["-q", STRUCT_PFM, "-u", "-m", "-10", str(d)], engine=OracleEngine(), out=o2)
    rows = lambda t: sorted(l.split("\t")[:-1] for l in t.splitlines()[1:])      # Match_ID depends on glob order
    assert rows(o1.getvalue()) == rows(o2.getvalue()) and len(rows(o1.getvalue())) > 5


def _write_multi_pfm(path, motifs):
    """writer of the reference's multi-PFM format (pfmutil.py:115-133)"""
    with open(path, "w") as f:
        for mid, letters, M in motifs:
            f.write("#" + mid + "\n#PO" + "".join("\t" + l for l in letters) + "\n")
            for i, row in enumerate(M):
                f.write(str(i) + "".join("\t" + str(float(x)) for x in row) + "\n")
            f.write("\n")


def test_multi_pfm_library_equals_one_scan_per_motif(tmp_path):
    """N1: a multi-PFM library scanned in one go == scanning each PFM on its own"""
    rng = np.random.default_rng(21)
    motifs = [("M%03d" % k, list("ACGU"), rng.dirichlet(np.full(4, 0.5), size=int(rng.integers(6, 13)))) for k in range(5)]
    lib = str(tmp_path / "lib.pfm")
    _write_multi_pfm(lib, motifs)
    assert pssm.is_multi_pfm(lib) and not pssm.is_multi_pfm(SEQ_PFM)
    got = list(pssm.read_multi_pfm(lib))
    assert [g[0] for g in got] == [m[0] for m in motifs]
    for (mid, counts), (_, letters, M) in zip(got, motifs):
        assert list(counts.keys()) == letters and np.array_equal(np.stack([counts[l] for l in letters], 1), M)
    P = pssm.load_pssms(lib, 0.01, fasta.RNA, None)
    assert list(P.keys()) == [m[0] for m in motifs]
    recs = [fasta.Record("r%d" % i, "r%d" % i, "".join(rng.choice(list("ACGT"), size=int(rng.integers(20, 300)))))
            for i in range(12)]
    eng = OracleEngine()
    both = scanner.scan_records(eng, recs, P, fasta.RNA, 1.0)
    singles = pd.concat([scanner.scan_records(eng, recs, {k: v}, fasta.RNA, 1.0) for k, v in P.items()], ignore_index=True)
    assert len(both) == len(singles) > 20
    key = ["Sequence_ID", "Start", "Motif_ID"]
    rid = {r.id: i for i, r in enumerate(recs)}
    singles = singles.assign(_r=singles["Sequence_ID"].map(rid)).sort_values(["_r", "Start", "Motif_ID"], kind="stable")
    pd.testing.assert_frame_equal(both.reset_index(drop=True), singles.drop(columns="_r").reset_index(drop=True))
    # per record the rows are ordered like sort_values(['Start', 'Motif_ID']) (rnascan.py:286)
    for _, g in both.groupby("Sequence_ID", sort=False):
        assert g[["Start", "Motif_ID"]].values.tolist() == sorted(g[["Start", "Motif_ID"]].values.tolist())
    out = io.StringIO()
    fa = tmp_path / "r.fa"
    fa.write_text("".join(">%s\n%s\n" % (r.id, r.seq) for r in recs))
    cli.main(["-p", lib, "-u", "-C", "0.01", "-m", "1", str(fa)], engine=eng, out=out)
    assert len(out.getvalue().splitlines()) == len(both) + 1


def _library_inputs(tmp_path, n_pairs=18, seed=5):
    """two multi-PFM libraries (sequence + structure) sharing motif ids, mixed widths, and matching FASTA + profiles"""
    rng = np.random.default_rng(seed)
    seq_m, st_m = [], []
    for k in range(n_pairs):
        w = int(rng.choice([6, 8, 8, 12]))
        seq_m.append(("RBP%02d" % k, list("ACGU"), rng.dirichlet(np.full(4, 0.5), size=w)))
        st_m.append(("RBP%02d" % k, list("EHTBLRM"), rng.dirichlet(np.full(7, 0.5), size=w)))
    st_m.append(("STRUCT_ONLY", list("EHTBLRM"), rng.dirichlet(np.full(7, 0.5), size=8)))      # no partner: never reported
    lib_s, lib_t = str(tmp_path / "seq_lib.pfm"), str(tmp_path / "struct_lib.pfm")
    _write_multi_pfm(lib_s, seq_m)
    _write_multi_pfm(lib_t, st_m)
    d = tmp_path / "avgdir"
    d.mkdir()
    fa = tmp_path / "recs.fa"
    with open(fa, "w") as f:
        for i in range(14):
            L = int(rng.integers(0, 350))
            f.write(">t%d transcript %d\n%s\n" % (i, i, "".join(rng.choice(list("ACGT"), size=L))))
            with open(d / ("structure.t%d.txt" % i), "w") as g:
                g.write("PO\t" + "\t".join("BEHLMRT") + "\n")
                p = rng.dirichlet(np.full(7, 0.3), size=L) if L else np.zeros((0, 7))
                for j in range(L):
                    g.write(str(j) + "\t" + "\t".join(repr(float(x)) for x in p[j]) + "\n")
    return lib_s, lib_t, str(fa), str(d)


def test_combined_scan_over_every_library_pair(tmp_path):
    """N1 / config 5 productised: `-p seq_library -q struct_library fasta dir/` reports EVERY motif pair that shares an
    id (one Motif_ID.Seq / Motif_ID.Struct per row), equal to scanning each pair on its own with the reference's
    one-motif semantics (rnascan.py:262, :422-433) and merging the tables"""
    lib_s, lib_t, fa, d = _library_inputs(tmp_path)
    eng = OracleEngine()
    ps = pssm.load_pssms(lib_s, 0.01, fasta.RNA, None)
    pt = pssm.load_pssms(lib_t, 0.01, fasta.STRUCT, None)
    assert len(ps) == 18 and len(pt) == 19
    pairs = scanner.pair_motifs(ps, pt)
    assert pairs == sorted((k, k) for k in ps) and ("STRUCT_ONLY", "STRUCT_ONLY") not in pairs
    recs = list(fasta.parse_sequences(fa))
    named = scanner.load_profile_dir(d)
    lib = scanner.scan_combined(eng, recs, named, ps, pt, -9.0, "aligned", np.float64)
    singles = [scanner.scan_combined(eng, recs, named, {a: ps[a]}, {b: pt[b]}, -9.0, "aligned", np.float64) for a, b in pairs]
    want = pd.concat(singles, ignore_index=True)
    rid = {r.id: i for i, r in enumerate(recs)}
    want = want.assign(_r=want["Sequence_ID"].map(rid)).sort_values(["_r", "Start", "Motif_ID.Seq", "Motif_ID.Struct"], kind="stable")
    assert len(lib) == len(want) > 100 and lib["Motif_ID.Seq"].nunique() > 10
    pd.testing.assert_frame_equal(lib.reset_index(drop=True), want.drop(columns="_r").reset_index(drop=True))
    # one structure motif against a library (and the reverse) = what the join of the two tables gives
    one = {"RBP03": pt["RBP03"]}
    widths = {k for k in ps if ps[k].length == pt["RBP03"].length}
    got = scanner.scan_combined(eng, recs, named, ps, one, -9.0, "aligned", np.float64)
    assert set(got["Motif_ID.Seq"]) <= widths and (got["Motif_ID.Struct"] == "RBP03").all() and len(got) > 20
    join = scanner.combine(scanner.scan_records(eng, recs, ps, fasta.RNA, -9.0), scanner.scan_profiles(eng, named, one, -9.0, "aligned", np.float64))
    assert len(join) == len(got)
    a = join[scanner.COMBINED_COLUMNS].sort_values(["Sequence_ID", "Start", "Motif_ID.Seq"]).reset_index(drop=True)
    b = got.sort_values(["Sequence_ID", "Start", "Motif_ID.Seq"]).reset_index(drop=True)
    pd.testing.assert_frame_equal(a, b, check_dtype=False)
    # through the command line
    out = io.StringIO()
    cli.main(["-p", lib_s, "-q", lib_t, "-u", "-C", "0.01", "-m", "-9", fa, d], engine=eng, out=out)
    assert len(out.getvalue().splitlines()) == len(lib) + 1
    # libraries of different size without a common id cannot be paired: the caller joins two tables
    other = {"X" + k: v for k, v in list(pt.items())[:5]}
    assert scanner.pair_motifs(ps, other) is None
    assert scanner.scan_combined(eng, recs, named, ps, other, -9.0) is None


@pytest.mark.parametrize("minscore", ["8", "-3"])
def test_native_ingest_path_and_record_path_write_the_same_bytes(tmp_path, monkeypatch, minscore):
    """a plain FASTA goes through pfmscan_fasta_encode / span columns, a gzipped one through Records and Python strings"""
    import gzip
    plain = str(tmp_path / "n.fa")
    nasty_fasta(plain)
    packed = plain + ".gz"
    with open(plain, "rb") as src, gzip.open(packed, "wb") as dst:
        dst.write(src.read())
    outs = []
    for path, batch in ((plain, None), (packed, None), (plain, "300")):
        if batch:
            monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", batch)        # several batches: Match_ID and buffers carry over
        out = io.StringIO()
        cli.main(["-p", SEQ_PFM, "-u", "-m", minscore, path], engine=OracleEngine(), out=out)
        outs.append(out.getvalue())
    assert outs[0] == outs[1] == outs[2]
    rows = outs[0].splitlines()
    assert len(rows) > (10 if minscore == "8" else 100)
    df = pd.read_csv(io.StringIO(outs[0]), sep="\t")
    assert df["Match_ID"].tolist() == list(range(1, len(df) + 1))
    recs = {r.id: r for r in fasta.parse_sequences(plain)}
    for r in df.itertuples():                                            # every row is what its record says
        rec = recs[r.Sequence_ID]
        assert r.Description == rec.description
        assert r.Sequence == fasta.preprocess_seq(rec.seq, True)[r.Start - 1:r.End]


def test_struct_library_one_pass_equals_per_motif_scans():
    """structure-only PFM library over profiles: the one-pass library path of _scan_profile_stream gives the table the
    per-motif loop gives (Motif_ID order inside a Start, windows that run over a record end dropped on the host)"""
    from collections import OrderedDict
    from engines import OracleEngine
    from rnascan_amd import pack, pssm, scanner
    rng = np.random.default_rng(21)
    named = []
    for i in range(11):
        L = int(rng.integers(3, 120))
        named.append(("r%d" % i, list("BEHLMRT"), rng.dirichlet(np.full(7, 0.3), size=L)))
    lib = OrderedDict()
    for k in (3, 1, 2, 0):                                   # ids out of order: the table sorts by Motif_ID inside a Start
        m = 6 if k < 3 else 9
        counts = rng.dirichlet(np.full(7, 0.5), size=m)
        d = OrderedDict((l, counts[:, c]) for c, l in enumerate(pack.STRUCT_COLUMNS))
        lib["m%d" % k] = pssm.PSSM(pack.STRUCT_COLUMNS, pssm.log_odds(pssm.normalize(d, 0.01), None))

    import pandas as pd
    eng = OracleEngine()
    one_pass = scanner.scan_profiles(eng, named, lib, -6.0, "aligned", np.float64)
    per_motif = scanner.scan_profiles(_without_library(eng), named, lib, -6.0, "aligned", np.float64)
    assert len(one_pass) > 30
    pd.testing.assert_frame_equal(one_pass.reset_index(drop=True), per_motif.reset_index(drop=True))
    # no window runs over a record end
    lengths = {n: p.shape[0] for n, _, p in named}
    assert all(e <= lengths[sid] for sid, e in zip(one_pass["Sequence_ID"], one_pass["End"]))


def _without_library(engine):
    """an engine object that offers no library_hits (forces the per-motif loop)"""
    class Plain(object):
        def __init__(self, e):
            self._e = e

        def scan(self, *a, **k):
            return self._e.scan(*a, **k)

        def hits(self, *a, **k):
            return self._e.hits(*a, **k)
    return Plain(engine)


def test_motif_loops_say_that_they_rescan_the_stream():
    """a loop over motifs must not take the one-shot pipeline per motif (it leaves nothing staged: every motif would upload
    the stream again); a single motif may"""
    from collections import OrderedDict
    from engines import OracleEngine
    from rnascan_amd import pack, pssm, scanner
    rng = np.random.default_rng(2)
    named = [("r%d" % i, list("BEHLMRT"), rng.dirichlet(np.full(7, 0.3), size=40)) for i in range(3)]

    def lib(widths):
        out = OrderedDict()
        for k, m in enumerate(widths):
            counts = rng.dirichlet(np.full(7, 0.5), size=m)
            d = OrderedDict((l, counts[:, c]) for c, l in enumerate(pack.STRUCT_COLUMNS))
            out["m%d" % k] = pssm.PSSM(pack.STRUCT_COLUMNS, pssm.log_odds(pssm.normalize(d, 0.01), None))
        return out

    seen = []

    class Spy(OracleEngine):
        def hits(self, *a, one_shot=True, **k):
            seen.append(one_shot)
            return OracleEngine.hits(self, *a, **k)

    scanner.scan_profiles(Spy(), named, lib([6]), -6.0)
    assert seen == [True]
    del seen[:]
    scanner.scan_profiles(Spy(), named, lib([6, 7, 8]), -6.0)          # three widths: three single-motif scans of one stream
    assert seen == [False, False, False]


def test_float32_store_under_float64_rows_is_scanned_in_place(tmp_path):
    """a float32 packed store whose records sit in FASTA order, under the default (float64) rows: the batch the engine gets
    IS the mapped file (no float64 copy of up to 32 x RNASCAN_BATCH_POSITIONS rows), and the table equals the float64-copy
    one -- widening float32 rows changes no score.  A float64 store under float32 rows is a real downcast: one copy."""
    from rnascan_amd import store
    rng = np.random.default_rng(8)
    d = tmp_path / "avg"
    d.mkdir()
    fa = tmp_path / "s.fa"
    with open(fa, "w") as f:
        for i in range(9):
            L = int(rng.integers(30, 200))
            f.write(">r%d\n%s\n" % (i, "".join(rng.choice(list("ACGU"), size=L))))
            prof = rng.dirichlet(np.full(7, 0.3), size=L)
            with open(d / ("structure.r%d.txt" % i), "w") as g:
                g.write("PO\t" + "\t".join("BEHLMRT") + "\n")
                for k, row in enumerate(prof):
                    g.write(str(k) + "\t" + "\t".join(str(float(x)) for x in row) + "\n")

    class Spy(OracleEngine):
        seen = []

        def hits(self, stream, *a, **k):
            Spy.seen.append(stream.profile)
            return OracleEngine.hits(self, stream, *a, **k)

    sp = {"s": pssm.pfm2pssm(SEQ_PFM, 0.01, fasta.RNA, None)}
    tp = {"t": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
    recs = fasta.LazyFasta(str(fa))
    tables = {}
    for sdtype in (np.float32, np.float64):
        sdir = str(tmp_path / ("store_%s" % np.dtype(sdtype).name))
        store.build_store(str(d), sdir, sdtype)
        ps = store.ProfileStore(sdir)
        order = [ps.ids.index(i) for i in recs.ids]
        assert order == sorted(order)                       # r0..r8 sort like the FASTA
        for rows in (np.float32, np.float64):
            Spy.seen = []
            pre = (list(recs.ids), ps.letters, ps.stream(0, len(ps.ids)))
            df = scanner.scan_combined(Spy(), recs[0:len(recs)], None, sp, tp, -25.0, "aligned", rows, prepacked=pre)
            assert len(df) > 5 and len(Spy.seen) == 1
            in_place = np.shares_memory(Spy.seen[0], ps.profile)
            assert in_place == (not (sdtype is np.float64 and rows is np.float32)), (sdtype, rows)
            tables[(sdtype, rows)] = df
    pd.testing.assert_frame_equal(tables[(np.float32, np.float32)], tables[(np.float32, np.float64)])
    pd.testing.assert_frame_equal(tables[(np.float32, np.float32)], tables[(np.float64, np.float32)])


class _OracleLetterLibraries(OracleEngine):
    """OracleEngine + the letter-library entry point of HipEngine (per-motif oracle scans merged into the (position, motif) order the
    device returns): lets the scanner's letter-library branches run on a machine without a GPU"""

    def library_hits_letters(self, stream, seq_tables, struct_tables, thr_seq, thr_struct):
        from oracle import oracle
        n = struct_tables.shape[0]
        tt = np.broadcast_to(np.asarray(thr_struct, dtype=np.float64), (n,))
        ts = None if seq_tables is None else np.broadcast_to(np.asarray(thr_seq, dtype=np.float64), (n,))
        pos, mo, sq_l, st_l = [], [], [], []
        for k in range(n):
            if seq_tables is None:
                st = oracle.stream_letters_f64(stream.codes, struct_tables[k])
                p = oracle.stream_hits(None, st, -np.inf, tt[k])
                sq = None
            else:
                sq = oracle.stream_seq(stream.codes, seq_tables[k])
                st = oracle.stream_letters_f64(stream.codes2, struct_tables[k])
                p = oracle.stream_hits(sq, st, ts[k], tt[k])
            pos.append(p)
            mo.append(np.full(p.size, k, dtype=np.int32))
            st_l.append(st[p])
            sq_l.append(np.zeros(p.size, dtype=np.float32) if sq is None else sq[p])
        pos, mo, sq_l, st_l = np.concatenate(pos), np.concatenate(mo), np.concatenate(sq_l), np.concatenate(st_l)
        order = np.lexsort((mo, pos))
        return pos[order], mo[order], (None if seq_tables is None else sq_l[order]), st_l[order]


def test_letter_library_branches_of_the_scanner_equal_the_per_motif_loops(tmp_path):
    """`-q struct_library structs.fa` and `-p seq_library -q struct_library seqs.fa structs.fa` (SURVEY 8f N1 x N4): with an engine
    that offers library_hits_letters the scanner makes ONE call per PFM width; the tables are those of the per-motif / per-pair
    loops (the reference's shape: one motif per side, rnascan.py:262) byte for byte"""
    rng = np.random.default_rng(77)
    seq_m, st_m = [], []
    for k in range(9):
        w = int(rng.choice([6, 6, 9, 9, 9, 33]))           # 33: too wide for the structure-letter library kernel -> its own scan
        seq_m.append(("RBP%02d" % k, list("ACGU"), rng.dirichlet(np.full(4, 0.5), size=w)))
        st_m.append(("RBP%02d" % k, list("EHTBLRM"), rng.dirichlet(np.full(7, 0.5), size=w)))
    lib_s, lib_t = str(tmp_path / "s.pfm"), str(tmp_path / "t.pfm")
    _write_multi_pfm(lib_s, seq_m)
    _write_multi_pfm(lib_t, st_m)
    with open(tmp_path / "s.fa", "w") as f, open(tmp_path / "t.fa", "w") as g:
        for i in range(25):
            L = int(rng.integers(0, 400))
            f.write(">r%d a\n%s\n" % (i, "".join(rng.choice(list("ACGUN"), size=L, p=[.24, .24, .24, .24, .04]))))
            g.write(">r%d b\n%s\n" % (i, "".join(rng.choice(list("EHTBLRMehtx"), size=L))))
    calls = []

    class Counting(_OracleLetterLibraries):
        def library_hits_letters(self, *a, **k):
            calls.append(a[2].shape)
            return _OracleLetterLibraries.library_hits_letters(self, *a, **k)

    for argv in (["-q", lib_t, "-u", "-m", "1", str(tmp_path / "t.fa")],
                 ["-p", lib_s, "-q", lib_t, "-u", "-m", "-1", str(tmp_path / "s.fa"), str(tmp_path / "t.fa")]):
        del calls[:]
        got, want = io.StringIO(), io.StringIO()
        cli.main(argv, engine=Counting(), out=got)
        cli.main(argv, engine=OracleEngine(), out=want)
        assert got.getvalue() == want.getvalue() and want.getvalue().count("\n") > 30
        assert len(calls) >= 2 and all(len(shape) == 3 and shape[0] > 1 for shape in calls)      # one call per width, several motifs each
