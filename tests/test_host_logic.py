"""Host-side logic (no GPU): the known answers the reference's own tests hold
(tests/motif_scan_test.py, tests/preprocess_seq_test.py), PFM -> PSSM, packing."""
import bz2
import gzip
import math
import os
import shutil

import numpy as np
import pytest

from conftest import DATA_DIR
from rnascan_amd import fasta, pack, pssm, shard


# ---- the reference's own unit tests, restated ---------------------------------
def test_parse_sequences_reference_fixture():
    recs = list(fasta.parse_sequences([os.path.join(DATA_DIR, "test.fa")]))
    assert [r.id for r in recs] == ["read1", "read2"]
    assert all(r.seq == "UUUUGCUCUGUAUAUA" for r in recs)


def test_compute_background_reference_known_answer(golden):
    g = golden["ref_tests"]["compute_background"]
    bg = fasta.compute_background([os.path.join(DATA_DIR, g["fasta"])], fasta.RNA, verbose=False)
    for key, value in g["expected"].items():
        assert round(abs(bg[key] - value), g["places"]) == 0      # assertAlmostEqual(..., 3)
    assert bg == {"G": 5 / 36, "A": 7 / 36, "U": 19 / 36, "C": 5 / 36}
    assert list(bg.keys()) == list("GAUC")


def test_preprocess_seq_reference_cases(golden):
    for c in golden["ref_tests"]["preprocess_seq"]:
        got = fasta.preprocess_seq(c["seq"], target_is_rna=(c["target"] == "RNA"), source_is_rna=(c["source"] == "RNA"))
        assert got == c["expected"], c


def test_compute_background_matches_oracle_loop(oracle):
    recs = list(fasta.parse_sequences(os.path.join(DATA_DIR, "HIST2H3C_3p_end.fa")))
    s = fasta.preprocess_seq(recs[0].seq, True)
    want = oracle.compute_background([s], "GAUC")
    got = fasta.compute_background(os.path.join(DATA_DIR, "HIST2H3C_3p_end.fa"), fasta.RNA, verbose=False)
    assert got == want


# ---- FASTA details ---------------------------------------------------------------
def test_fasta_header_multiline_and_compressed(tmp_path):
    src = os.path.join(DATA_DIR, "HIST2H3C_3p_end.fa")
    rec = list(fasta.parse_sequences(src))[0]
    assert rec.id == "hg19_dna"
    assert rec.description.startswith("hg19_dna range=chr1:149824452-149824687")
    assert len(rec.seq) == 236 and set(rec.seq) <= set("ACGT")
    gz, bz = str(tmp_path / "a.fa.gz"), str(tmp_path / "a.fa.bz2")
    with open(src, "rb") as f, gzip.open(gz, "wb") as g:
        shutil.copyfileobj(f, g)
    with open(src, "rb") as f, bz2.open(bz, "wb") as g:
        shutil.copyfileobj(f, g)
    assert list(fasta.parse_sequences([gz, bz])) == [rec, rec]


def test_load_background_literal_file(tmp_path):
    p = tmp_path / "bg.txt"
    p.write_text("{'A': 0.3, 'C': 0.2, 'G': 0.2, 'U': 0.3}")
    assert fasta.load_background(str(p), False, None, fasta.RNA)["A"] == 0.3
    assert fasta.load_background(None, True, None, fasta.RNA) is None


def test_read_profile_and_listing(tmp_path):
    src = os.path.join(DATA_DIR, "HIST2H3C_3p_end_structure.txt")
    letters, prof = fasta.read_profile(src)
    assert letters == list("BEHLMRT") and prof.shape == (236, 7)
    assert prof[0].tolist() == [0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    shutil.copyfile(src, tmp_path / "structure.hg19_dna.txt")
    shutil.copyfile(src, tmp_path / "structure.a.b.txt")
    (tmp_path / "other.txt").write_text("x")
    assert sorted(fasta.list_profiles(str(tmp_path))) == sorted(
        [("hg19_dna", str(tmp_path / "structure.hg19_dna.txt")), ("a.b", str(tmp_path / "structure.a.b.txt"))])


# ---- PFM -> PSSM -------------------------------------------------------------------
def test_pfm2pssm_matches_oracle_loops_and_golden(oracle, golden):
    path = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt")
    P = pssm.pfm2pssm(path, 0.0, fasta.RNA, None)
    assert list(P.keys()) == list("GAUC") and P.length == 18
    raw = pssm.read_pfm(path)
    counts = {l: raw[l].tolist() for l in "GAUC"}          # alphabet.letters order, as Biopython sums it
    want = oracle.log_odds(oracle.normalize(counts, 0.0), None)
    for l in "ACGU":
        assert P[l].tolist() == want[l]
    # the operand recorded next to the reference's outputs in the fixture
    g = golden["calculate_route"]["pssm"]
    for l in "ACGU":
        assert P[l].tolist() == g[l]
    assert P.letter_table("ACGU")[0, :4].tolist() == [1.579627261360602, -1.9358833760604863, -1.1421163018076363, -1.75332079856928]
    assert np.isnan(P.letter_table("ACGU")[:, 4:]).all()


def test_pfm2pssm_struct_inf_cells_and_background():
    path = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt")
    P = pssm.pfm2pssm(path, 0.0, fasta.STRUCT, None)
    M = P.matrix("BEHLMRT")
    assert np.isneginf(M).sum() == 21 and np.isneginf(M[1]).sum() == 5
    bg = {l: 1.0 for l in fasta.STRUCT}                                       # unnormalised background is renormalised
    assert np.array_equal(pssm.pfm2pssm(path, 0.0, fasta.STRUCT, bg).matrix("BEHLMRT"), M)
    P1 = pssm.pfm2pssm(path, 0.01, fasta.STRUCT, None)
    assert np.isfinite(P1.matrix("BEHLMRT")).all()


def test_log_odds_special_cells():
    pw = {"A": np.array([0.5, 0.0]), "C": np.array([0.5, 1.0])}
    lo = pssm.log_odds(pw, {"A": 0.0, "C": 1.0})
    assert lo["A"][0] == math.inf and math.isnan(lo["A"][1])
    assert lo["C"].tolist() == [-1.0, 0.0]
    lo = pssm.log_odds(pw, None)
    assert lo["A"].tolist() == [0.0, -math.inf]


def test_wrong_alphabet_raises_keyerror():
    with pytest.raises(KeyError):
        pssm.pfm2pssm(os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt"), 0.0, fasta.STRUCT, None)


def test_test_pfm_header_order_is_honoured():
    """tests/test_seq_pfm.txt has header order G A U C: columns are matched by letter"""
    P = pssm.pfm2pssm(os.path.join(DATA_DIR, "test_seq_pfm.txt"), 0.0, fasta.RNA, None)
    raw = pssm.read_pfm(os.path.join(DATA_DIR, "test_seq_pfm.txt"))
    assert list(raw.keys()) == list("GAUC")
    row0 = np.array([raw[l][0] for l in "GAUC"])
    assert np.allclose(P.letter_table("ACGU")[0, :4], np.log2(np.array([raw[l][0] for l in "ACGU"]) / row0.sum() / 0.25))


# ---- packing -----------------------------------------------------------------------
def test_encode_rna_letter_classes():
    assert pack.encode_rna("ACGUTacgutN-").tolist() == [0, 1, 2, 3, 3, 0, 1, 2, 3, 3, 7, 7]
    assert pack.encode_letters("EhtBLRMx.", pack.STRUCT_LETTERS).tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 7]


def test_pack_layout_and_locate():
    codes = [np.array([0, 1, 2], np.uint8), np.zeros(0, np.uint8), np.array([3], np.uint8)]
    profs = [np.full((3, 7), 0.5), np.zeros((0, 7)), np.ones((1, 7))]
    s = pack.pack(codes, profs)
    assert s.codes.tolist() == [0, 1, 2, 7, 7, 3, 7]
    assert s.offsets.tolist() == [0, 4, 5] and s.lengths.tolist() == [3, 0, 1] and s.n_pos == 7
    assert s.profile.dtype == np.float32 and s.profile.shape == (7, 7)
    assert s.profile[3].tolist() == [0] * 7 and s.profile[5].tolist() == [1] * 7
    rec, start = s.locate([0, 2, 5])
    assert rec.tolist() == [0, 0, 2] and start.tolist() == [0, 2, 0]
    assert s.n_windows(2) == 2 and s.window_mask(2).tolist() == [True, True, False, False, False, False, False]
    with pytest.raises(ValueError):
        pack.pack(codes, [np.zeros((2, 7))] * 3)


# ---- sharding ------------------------------------------------------------------------
def test_partition_contiguous_balanced():
    L = [3000] * 10
    assert shard.partition(L, 1) == [(0, 10)]
    assert shard.partition(L, 2) == [(0, 5), (5, 10)]
    parts = shard.partition([10, 10, 10, 1000, 10, 10], 3)
    assert parts[0][0] == 0 and parts[-1][1] == 6
    assert all(parts[i][1] == parts[i + 1][0] for i in range(2))
    assert shard.partition([], 4) == [(0, 0)] * 4
    assert shard.partition([5], 3)[-1][1] == 1
    rng = np.random.default_rng(0)
    L = rng.integers(1, 5000, size=1000)
    for w in (2, 4, 8):
        parts = shard.partition(L, w)
        loads = [int((L[a:b] + 1).sum()) for a, b in parts]
        assert sum(loads) == int((L + 1).sum())
        assert max(loads) - min(loads) <= 2 * 5001


# ---- streaming TSV writer (N3) --------------------------------------------------------
def test_tsv_writer_matches_pandas_to_csv(golden):
    import io
    import pandas as pd
    from rnascan_amd import table
    rng = np.random.default_rng(1)
    n = 5000
    f32 = np.round(rng.normal(0, 10, n).astype(np.float32), 3)
    f64 = rng.normal(0, 30, n)
    f64[:5] = [np.inf, -np.inf, np.nan, 1e-5, -1.7976931348623157e308]
    f32[5:8] = [np.float32(1e-5), np.float32(123456.789), np.float32(-0.0)]
    df = pd.DataFrame({"Sequence_ID": ["id%d" % (i // 7) for i in range(n)], "Description": "some text here",
                       "Motif_ID": "M1", "Start": np.arange(1, n + 1), "End": np.arange(12, n + 12),
                       "Sequence": ["ACGU"] * n, "LogOdds.Seq": f32, "LogOdds.Struct": f64,
                       "LogOdds.SeqStruct": f32.astype(np.float64) + f64})
    want = df.copy()
    want["Match_ID"] = list(range(1, n + 1))
    out = io.StringIO()
    assert table.write_frame(out, df, chunk=777) == n
    assert out.getvalue() == want.to_csv(sep="\t", index=False)
    # and on the reference's own captured table
    g = golden["combine"]
    gdf = pd.read_csv(io.StringIO(g["tsv"]), sep="\t", dtype={"LogOdds.Seq": np.float32}, keep_default_na=False)
    gdf["Description.Struct"] = ""
    out = io.StringIO()
    table.write_frame(out, gdf.drop(columns="Match_ID"))
    assert out.getvalue() == g["tsv"]


# ---- property-based host invariants -----------------------------------------------------
def test_pack_locate_partition_properties():
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=200, deadline=None, derandomize=True)
    @given(st.lists(st.integers(0, 50), min_size=1, max_size=30), st.integers(1, 12), st.integers(1, 9))
    def run(lengths, m, world):
        codes = [np.full(L, i % 4, dtype=np.uint8) for i, L in enumerate(lengths)]
        s = pack.pack(codes)
        assert s.n_pos == sum(lengths) + len(lengths)
        assert (s.codes[s.offsets + s.lengths] == pack.SEP).all()           # one separator after every record
        mask = s.window_mask(m)
        assert mask.sum() == s.n_windows(m) == sum(max(L - m + 1, 0) for L in lengths)
        pos = np.flatnonzero(mask)
        rec, start = s.locate(pos)
        assert ((start >= 0) & (start + m <= s.lengths[rec])).all()
        assert (s.offsets[rec] + start == pos).all()
        parts = shard.partition(lengths, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(lengths)
        assert all(a <= b for a, b in parts) and all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))

    run()


def test_tsv_writer_quotes_like_to_csv():
    """Description is the whole FASTA header: a tab, a double quote or a line break in it must be quoted exactly as
    DataFrame.to_csv(sep='\\t') does (csv.QUOTE_MINIMAL, rnascan.py:559-567)"""
    import io
    import pandas as pd
    from rnascan_amd import table
    df = pd.DataFrame({
        "Sequence_ID": ["a", "b", "c", "d"],
        "Description": ['a desc\twith "tab"', 'plain', 'say "hi"', "two\nlines"],
        "Motif_ID": ["m"] * 4, "Start": [1, 2, 3, 4], "End": [8, 9, 10, 11],
        "Sequence": ["ACGUACGU"] * 4, "LogOdds": np.array([1.5, 2.25, np.nan, 7.125], dtype=np.float32)})
    want = io.StringIO()
    ref = df.copy()
    ref["Match_ID"] = list(range(1, len(ref) + 1))
    ref.to_csv(want, sep="\t", index=False)
    got = io.StringIO()
    table.write_frame(got, df, match_id=True)
    assert got.getvalue() == want.getvalue()


def test_cli_says_why_a_pfm_wider_than_the_kernels_is_refused(tmp_path, capsys):
    """the reference's loops take any PFM width (_pwm.c:34-68); this build stops at PFMSCAN_MAX_WIDTH with a clear message
    (widths above PFMSCAN_MAX_M = 64 run the plain kernel: tests/test_gpu_wide.py)"""
    import pytest
    from rnascan_amd import _lib, cli, fasta
    pfm = tmp_path / "wide.pfm"
    with open(pfm, "w") as f:
        f.write("PO\tA\tC\tG\tU\n")
        for i in range(_lib.MAX_WIDTH + 1):
            f.write("%d\t0.25\t0.25\t0.25\t0.25\n" % i)
    with pytest.raises(SystemExit) as e:
        cli.load_motif(str(pfm), 0.01, fasta.RNA, None)
    assert e.value.code == 1
    err = capsys.readouterr().err
    assert "%d positions wide" % (_lib.MAX_WIDTH + 1) in err and "at most %d" % _lib.MAX_WIDTH in err


def test_round_decimals_is_pythons_round():
    """rnascan.py:273 rounds every reported score with Python's round(); for the fp64 structure-letter scores that is
    float.__round__ (nearest decimal of the exact value), which numpy.round is not"""
    from rnascan_amd import _lib
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 20, 50000), [2.6745, 1.0005, 0.0005, -0.0005, 2.675, 1e300, -1e-300, 0.0, -0.0, np.inf, np.nan,
                                                   -np.inf, 1.2345e15, 5e-4, 1.5e-3, 2.5e-3, 4503599627370495.5, 1e22],
                        np.round(rng.normal(0, 5, 20000), 3) + 0.0005, (rng.integers(-10 ** 7, 10 ** 7, 20000) * 2 + 1) / 2000.0])
    for nd in (3, 0, 1, 6):
        got = _lib.round_decimals(x, nd)
        want = np.array([round(float(v), nd) for v in x])
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (nd, x[~same][:5])
        assert np.array_equal(np.signbit(got), np.signbit(want))
    assert int((np.round(x, 3) != np.array([round(float(v), 3) for v in x])).sum()) > 100     # the two functions do differ
    assert _lib.round_decimals(np.zeros(0)).size == 0
    with pytest.raises(ValueError):
        _lib.round_decimals(x, 16)


def test_letter_lut_keeps_the_case_in_bit_3():
    from rnascan_amd import pack
    lut = pack.letter_lut("EHTBLRM", keep_case=True)
    assert [int(lut[ord(c)]) for c in "EHTBLRM"] == list(range(7))
    assert [int(lut[ord(c)]) for c in "ehtblrm"] == [k | pack.CASE_BIT for k in range(7)]
    assert lut[ord("X")] == pack.SEP and lut[ord("x")] == pack.SEP and lut[0] == pack.SEP
    plain = pack.letter_lut("EHTBLRM")
    assert np.array_equal(plain, np.where(lut == pack.SEP, pack.SEP, lut & 7))
    codes = pack.encode_letters("EeLlXm", "EHTBLRM", keep_case=True)
    assert codes.tolist() == [0, 8, 4, 12, 7, 14]


def _struct_scores(prof, P):
    """rnascan.py:302-307 in numpy: per-row dot, nan_to_num, sequential fp64 sum"""
    m = P.shape[0]
    n = prof.shape[0] - m + 1
    out = np.zeros(n)
    with np.errstate(invalid="ignore", over="ignore"):
        for j in range(m):
            out = out + np.nan_to_num((prof[j:j + n].astype(np.float64) * P[j]).sum(axis=1))
    return out


def test_float32_storage_bound_holds_against_brute_force():
    """scanner.float32_storage_bound is a PROVEN bound: on random structure PFMs (with -inf cells too) and profiles whose
    rows sum to 1 (exact zeros, entries just above powers of two -- the worst case for a relative rounding error) the
    scores from float32-rounded rows never differ by more than it from the float64 ones"""
    from collections import OrderedDict
    from rnascan_amd import scanner
    rng = np.random.default_rng(17)
    worst_ratio = 0.0
    for trial in range(40):
        m = int(rng.integers(1, 24))
        counts = rng.dirichlet(np.full(7, rng.choice([0.3, 1.0, 5.0])), size=m)
        if trial % 3 == 0:
            counts[rng.random((m, 7)) < 0.15] = 0.0
        pc = float(rng.choice([0.0, 0.01, 0.5]))
        pm = pssm.PSSM(pack.STRUCT_LETTERS, pssm.log_odds(pssm.normalize(OrderedDict((l, counts[:, k]) for k, l in enumerate(pack.STRUCT_LETTERS)), pc), None))
        bound = scanner.float32_storage_bound({"x": pm})
        P = pm.matrix(list(pack.STRUCT_LETTERS))
        prof = rng.dirichlet(np.full(7, 0.3), size=4000)
        prof[prof < 0.02] = 0.0
        prof[::7] = np.nextafter(np.float32(0.5), np.float32(1.0)) * np.eye(7)[rng.integers(0, 7, size=prof[::7].shape[0])]   # p just above 1/2
        prof /= np.maximum(prof.sum(axis=1, keepdims=True), 1.0)                    # rows sum to at most 1
        a, b = _struct_scores(prof, P), _struct_scores(prof.astype(np.float32), P)
        ok = np.isfinite(a) & (np.abs(a) < 1e300)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~ok & ~np.isnan(a)], b[~ok & ~np.isnan(a)])
        err = np.abs(a[ok] - b[ok]).max(initial=0.0)
        assert err <= bound * (1 + 1e-9) + 1e-13, (trial, err, bound)
        worst_ratio = max(worst_ratio, err / bound if bound else 0.0)
    assert worst_ratio > 0.02           # and it is not absurdly loose: some case comes within 50x of it


def test_profile_dtype_is_picked_from_the_pfm(capsys):
    """auto: float32 rows only when the structure PFM proves the storage error below 5e-7; the reference's own w = 18
    example does not (its measured float32 error is 8.6e-7), a flat PFM does; the flag still overrides"""
    import argparse
    from collections import OrderedDict
    from rnascan_amd import cli, scanner
    slbp = pssm.load_pssms(os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt"), 0.0, fasta.STRUCT, None)
    t, bound = scanner.pick_profile_dtype("auto", slbp)
    assert t is np.float64 and bound > 8.6e-7              # the bound covers what tests/test_gpu_parity.py measures on that example
    flat = OrderedDict((l, np.full(6, 1.0 / 7) + 0.004 * k) for k, l in enumerate(pack.STRUCT_LETTERS))
    pm = {"flat": pssm.PSSM(pack.STRUCT_LETTERS, pssm.log_odds(pssm.normalize(flat, 0.0), None))}
    t, bound = scanner.pick_profile_dtype("auto", pm)
    assert t is np.float32 and bound < 5e-7
    assert scanner.pick_profile_dtype("float64", pm)[0] is np.float64 and scanner.pick_profile_dtype("float32", slbp)[0] is np.float32
    args = argparse.Namespace(profile_dtype="auto")
    assert cli.profile_type(args, slbp) is np.float64 and cli.profile_type(args, slbp) is np.float64
    err = capsys.readouterr().err
    assert err.count("stored as float64") == 1 and "could cost up to" in err      # announced once
