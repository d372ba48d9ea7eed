"""The reference's call signatures (rnascan_amd/compat.py, PSSM.search / PSSM.calculate): SURVEY 8b rows (i)-(iv).
Host logic on the TEST-ONLY oracle engine here; the same checks run against the goldens on the GPU
(tests/test_gpu_cli.py::test_compat_*)."""
import argparse
import os
import shutil

import numpy as np
import pytest

from conftest import DATA_DIR
from engines import OracleEngine
from rnascan_amd import compat, fasta, pssm

SEQ_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt")
STRUCT_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt")
HIST_FA = os.path.join(DATA_DIR, "HIST2H3C_3p_end.fa")
HIST_PROFILE = os.path.join(DATA_DIR, "HIST2H3C_3p_end_structure.txt")


def golden_pssm(case, letters):
    """PSSM object from a golden case's recorded operand (rows x letters in the recorded order)"""
    P = np.array(case["pssm"], dtype=np.float64)
    order = case["pssm_letter_order"]
    return pssm.PSSM(letters, {l: P[:, order.index(l)] for l in letters})


def check_compat(engine, golden):
    compat.set_default_engine(engine)
    try:
        # (ii) calculate: the reference's own calculate() output (matrix.py:68-81)
        g = golden["calculate_route"]
        pm = pssm.PSSM("GAUC", {l: np.array(g["pssm"][l]) for l in "GAUC"})
        got = pm.calculate(g["sequence"])
        assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), np.array(g["scores"], dtype=np.float32).view(np.uint32))
        one = pm.calculate(g["single_window_sequence"])
        assert np.ndim(one) == 0 and float(one) == g["single_window_score"]
        # (i) search: (position, score) of every window with score > threshold, strict, window order
        want = np.array(g["scores"], dtype=np.float32)
        for thr in (-1e30, float(np.sort(want[np.isfinite(want)])[len(want) // 2]), float("-inf")):
            hits = list(pm.search(g["sequence"].lower(), threshold=thr, both=False))      # search upper-cases
            exp = [(i, x) for i, x in enumerate(want) if float(x) > thr]
            assert [p for p, _ in hits] == [p for p, _ in exp]
            assert all(np.float32(a) == np.float32(b) for (_, a), (_, b) in zip(hits, exp))
        with pytest.raises(ValueError):
            list(pm.search("ACGU", both=True))
        # generic alphabet: _py_calculate route (list of Python floats, NaN on an unknown letter)
        for case in golden["py_calculate"]:
            letters = case["letters"]
            T = np.array(case["table"], dtype=np.float64)[: case["m"]]
            sp = pssm.PSSM(letters, {l: T[:, k] for k, l in enumerate(letters)})
            got = sp.calculate(case["sequence"])
            got = [got] if np.ndim(got) == 0 else got
            want = case["scores"]
            assert len(got) == len(want)
            assert all((np.isnan(a) and np.isnan(b)) or a == b for a, b in zip(got, want))
        # (iv) scan_averaged_structure(struct_file, pssm, minscore): the reference's rows, both column pairings
        for case in golden["scan_averaged_structure"]:
            if not case["profile_file"]:
                continue
            pm_s = golden_pssm(case, fasta.STRUCT)
            df = compat.scan_averaged_structure(os.path.join(DATA_DIR, case["profile_file"]), {"m1": pm_s}, case["minscore"],
                                                pairing=case["pairing"])
            assert list(df.columns) == ["Motif_ID", "Start", "End", "Sequence", "LogOdds"]
            want = case["rows"]
            assert df["Start"].tolist() == [r[0] for r in want] and df["End"].tolist() == [r[1] for r in want]
            assert (df["Sequence"] == case["sequence_field"]).all()
            assert np.allclose(df["LogOdds"].to_numpy(), [r[2] for r in want], rtol=0, atol=1e-6)
    finally:
        compat.set_default_engine(None)


def test_compat_signatures_on_the_oracle_engine(golden):
    check_compat(OracleEngine(), golden)


def test_compat_scan_main_and_combine(golden, tmp_path):
    """scan_main(fasta | directory, pssm, alphabet, bg, args) + combine == the reference's combined table"""
    d = tmp_path / "avg"
    d.mkdir()
    shutil.copyfile(HIST_PROFILE, d / "structure.hg19_dna.txt")
    compat.set_default_engine(OracleEngine())
    try:
        bg = fasta.compute_background(HIST_FA, fasta.RNA, verbose=False)
        ps = {"SLBP_seq": pssm.pfm2pssm(SEQ_PFM, 0.01, fasta.RNA, bg)}
        pt = {"SLBP_struct": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
        args = argparse.Namespace(minscore=0.0, debug=False, cores=2)
        seq = compat.scan_main(HIST_FA, ps, "GAUC", bg, args)
        st = compat.scan_main(str(d), pt, "EHTBLRM", None, args)
        both = compat.combine(seq, st)
        compat._add_match_id(both)
        g = golden["combine"]
        assert list(both.columns) == g["columns"] and len(both) == len(g["rows"])
        for a, b in zip(both.itertuples(index=False), g["rows"]):
            for x, y in zip(a, b):
                if isinstance(y, float):
                    assert abs(float(x) - y) <= 1e-6
                else:
                    assert x == y
        # scan / scan_all with the reference's argument order
        rec = list(fasta.parse_sequences(HIST_FA))[0]
        rows = compat.scan(ps, fasta.preprocess_seq(rec.seq, True), "GAUC", 0.0)
        assert [r[1] for r in rows] == seq["Start"].tolist() and [r[4] for r in rows] == seq["LogOdds"].tolist()
        assert compat.scan_all(rec, ps, "GAUC", 0.0)["Start"].tolist() == seq["Start"].tolist()
    finally:
        compat.set_default_engine(None)
