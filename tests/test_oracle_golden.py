"""The CPU oracle against every golden vector captured from the reference
(tests/golden/make_golden.py) and, where oracle/_ref exists, against the
reference's own compiled _pwm.c on fresh random inputs."""
import os

import numpy as np
import pytest

from conftest import assert_f32_bits_equal, DATA_DIR


def test_pwm_goldens(oracle, golden):
    assert len(golden["pwm"]) >= 15
    for case in golden["pwm"]:
        M = np.array(case["matrix"], dtype=np.float64).reshape(-1, 4)
        got = oracle.pwm_calculate(case["sequence"], M)
        assert_f32_bits_equal(got, np.array(case["scores"], dtype=np.float32))


def test_pwm_survey_known_values(oracle, golden):
    """the numbers SURVEY.md 8(c) quotes for HIST2H3C x SLBP (-u -C 0)"""
    c = [c for c in golden["pwm"] if c["name"] == "hist_slbp_pc0_uniform"][0]
    s = c["scores"]
    assert len(s) == 219
    assert s[0] == -15.023512840270996 and s[1] == -14.387472152709961 and s[2] == -17.377042770385742
    assert s[211] == 4.357380390167236 and s[212] == 14.258893966674805 and s[213] == 3.640672445297241
    assert s[218] == -11.437420845031738
    assert c["sequence"][212:230] == "AAAGGCUCUUUUCAGAGC"
    assert c["matrix"][0] == [1.579627261360602, -1.9358833760604863, -1.1421163018076363, -1.75332079856928]


def test_oracle_vs_reference_build(oracle):
    ref = oracle.ref_pwm()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(123)
    letters = np.array(list("ACGUacgutTNX"))
    for m in (1, 3, 8, 12, 18, 40):
        for L in (m - 1, m, m + 1, 100, 3000):
            if L < 0:
                continue
            seq = "".join(rng.choice(letters, size=L))
            M = rng.normal(0, 5, size=(m, 4))
            if m > 3:
                M[1, 2] = -np.inf
            assert_f32_bits_equal(oracle.pwm_calculate(seq, M), ref.calculate(seq, M))


def test_py_calculate_goldens(oracle, golden):
    for case in golden["py_calculate"]:
        T = np.array(case["table"], dtype=np.float64)[: case["m"]]
        got = oracle.py_calculate(case["sequence"], case["letters"], T)
        want = np.array(case["scores"], dtype=np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    c = golden["py_calculate"][0]
    assert c["sequence"] == "EELLX" and c["scores"][:3] == [5.236304429672733, -2.681544712293105, -7.070899664657709]
    assert np.isnan(c["scores"][3])


def _profile_of(case):
    if case["profile_file"]:
        rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(DATA_DIR, case["profile_file"]))][1:]
        return np.array([[float(x) for x in r[1:]] for r in rows])
    return np.array(case["profile"], dtype=np.float64)


def test_scan_averaged_structure_goldens(oracle, golden):
    seen = set()
    for case in golden["scan_averaged_structure"]:
        prof = _profile_of(case)
        P = np.array(case["pssm"], dtype=np.float64)
        m = P.shape[0]
        got = oracle.scan_averaged_structure(prof, P)
        keep = got > case["minscore"]
        starts = np.flatnonzero(keep) + 1
        want = case["rows"]
        assert [r[0] for r in want] == starts.tolist()
        assert [r[1] for r in want] == (starts + m - 1).tolist()
        w = np.array([r[2] for r in want])
        g = got[keep]
        big = np.abs(w) > 1e9
        assert np.allclose(g[~big], w[~big], rtol=0, atol=1e-9)
        assert np.allclose(g[big], w[big], rtol=1e-12, atol=0)
        seen.add(case["pairing"])
    assert seen == {"positional", "aligned"}


def test_scan_averaged_structure_survey_values(golden):
    """positional (reference as run on py3) and label-aligned numbers"""
    cases = {c["name"]: c for c in golden["scan_averaged_structure"]}
    pos = {r[0]: r[2] for r in cases["hist_slbp_pc0_positional"]["rows"]}
    ali = {r[0]: r[2] for r in cases["hist_slbp_pc0_aligned"]["rows"]}
    assert pos[1] == -35.515272427965556 and pos[213] == -15.993903438970927
    assert ali[1] == -10.103018200227329 and ali[213] == 26.59971805529875 and ali[219] == -29.546474673279533


def test_stream_forms_match_record_forms(oracle):
    """the packed-stream oracle entry points are the same arithmetic as the per-record ones"""
    from rnascan_amd import pack
    rng = np.random.default_rng(4)
    m = 9
    seqs = ["".join(rng.choice(list("ACGUN"), size=L, p=[.24, .24, .24, .24, .04])) for L in (50, 3, 9, 200, 0, 31)]
    M = rng.normal(0, 2, size=(m, 4))
    T = np.full((m, 8), np.nan)
    T[:, :4] = M
    profs = [rng.dirichlet(np.full(7, 0.3), size=len(s)) if len(s) else np.zeros((0, 7)) for s in seqs]
    P = rng.normal(-1, 2, size=(m, 7))
    P[2, 3] = -np.inf
    s = pack.pack([pack.encode_rna(x) for x in seqs], profs, profile_dtype=np.float64)
    sq = oracle.stream_seq(s.codes, T)
    st = oracle.stream_struct(s.profile, P)
    for r, (seq, prof) in enumerate(zip(seqs, profs)):
        sl = s.record_slice(r, m)
        assert_f32_bits_equal(sq[sl], oracle.pwm_calculate(seq, M) if len(seq) >= m - 1 else np.zeros(0, np.float32))
        want = oracle.scan_averaged_structure(prof, P)
        assert np.array_equal(st[sl], want)
    assert np.isnan(sq[~s.window_mask(m)]).all()
