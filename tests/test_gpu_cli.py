"""End to end on the GPU: the drop-in CLI and the scanner layer with the real HipEngine,
against the reference's captured tables (config 1: example/ HIST2H3C + SLBP PFMs)."""
import io
import os
import shutil

import numpy as np
import pytest

from conftest import DATA_DIR

pytestmark = pytest.mark.gpu

SEQ_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt")
STRUCT_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt")
HIST_FA = os.path.join(DATA_DIR, "HIST2H3C_3p_end.fa")
HIST_PROFILE = os.path.join(DATA_DIR, "HIST2H3C_3p_end_structure.txt")


@pytest.fixture(scope="module")
def engine():
    from rnascan_amd import scanner
    e = scanner.HipEngine(0)
    yield e
    e.close()


@pytest.fixture()
def workdir(tmp_path):
    d = tmp_path / "avg"
    d.mkdir()
    shutil.copyfile(HIST_PROFILE, d / "structure.hg19_dna.txt")
    os.symlink(SEQ_PFM, tmp_path / "SLBP_seq.txt")
    os.symlink(STRUCT_PFM, tmp_path / "SLBP_struct.txt")
    (tmp_path / "bg_struct.txt").write_text(repr({l: 1.0 / 7 for l in "EHTBLRM"}))
    return tmp_path


def test_cli_combined_matches_reference_table(engine, golden, workdir):
    from rnascan_amd import cli
    from test_scanner_cpu import assert_tsv_equal
    out = io.StringIO()
    cli.main(["-p", str(workdir / "SLBP_seq.txt"), "-q", str(workdir / "SLBP_struct.txt"), "-C", "0.01", "-m", "0",
              "-B", str(workdir / "bg_struct.txt"), HIST_FA, str(workdir / "avg")], engine=engine, out=out)
    assert_tsv_equal(out.getvalue(), golden["combine"]["tsv"])
    # float32 profile storage (the benchmark's layout): same table within 1e-6
    out32 = io.StringIO()
    cli.main(["-p", str(workdir / "SLBP_seq.txt"), "-q", str(workdir / "SLBP_struct.txt"), "-C", "0.01", "-m", "0",
              "-B", str(workdir / "bg_struct.txt"), "--profile-dtype", "float32", HIST_FA, str(workdir / "avg")],
             engine=engine, out=out32)
    assert_tsv_equal(out32.getvalue(), golden["combine"]["tsv"], tol=1e-6)


def test_config1_minus_inf_all_219_windows(engine, golden):
    """BASELINE.json configs[0]: -m ' -inf' on the example inputs, both pairings"""
    from rnascan_amd import fasta, pssm, scanner
    rec = list(fasta.parse_sequences(HIST_FA))[0]
    sp = {"SLBP": pssm.pfm2pssm(SEQ_PFM, 0.0, fasta.RNA, None)}
    df = scanner.scan_records(engine, [rec], sp, fasta.RNA, float("-inf"))
    g = [c for c in golden["pwm"] if c["name"] == "hist_slbp_pc0_uniform"][0]
    assert len(df) == 219 and df["Start"].tolist() == list(range(1, 220))
    assert np.array_equal(df["LogOdds"].to_numpy(), np.round(np.array(g["scores"], dtype=np.float32), 3))
    assert df["Sequence"].iloc[212] == "AAAGGCUCUUUUCAGAGC" and float(df["LogOdds"].iloc[212]) == float(np.float32(14.259))
    tp = {"SLBP": pssm.pfm2pssm(STRUCT_PFM, 0.0, fasta.STRUCT, None)}
    cases = {c["name"]: c for c in golden["scan_averaged_structure"]}
    for pairing, name in (("positional", "hist_slbp_pc0_positional"), ("aligned", "hist_slbp_pc0_aligned")):
        d = scanner.scan_averaged_structure(engine, HIST_PROFILE, tp, float("-inf"), pairing)
        want = cases[name]["rows"]
        assert d["Start"].tolist() == [r[0] for r in want]
        assert np.abs(d["LogOdds"].to_numpy() - np.array([r[2] for r in want])).max() <= 1e-9


def test_scanner_layer_gpu_equals_oracle_engine(engine):
    """same tables from the HIP engine and the test-only oracle engine on ragged random records"""
    import pandas as pd
    from engines import OracleEngine
    from rnascan_amd import fasta, pssm, scanner
    rng = np.random.default_rng(11)
    recs, named = [], []
    for i in range(40):
        L = int(rng.integers(0, 900))
        recs.append(fasta.Record("id%d" % i, "id%d d" % i, "".join(rng.choice(list("ACGTNacgu"), size=L))))
        p = rng.dirichlet(np.full(7, 0.3), size=L) if L else np.zeros((0, 7))
        p[p < 0.02] = 0
        if i % 5:
            named.append(("id%d" % i, list("BEHLMRT"), p))
    sp = {"a": pssm.pfm2pssm(SEQ_PFM, 0.0, fasta.RNA, None)}
    tp = {"b": pssm.pfm2pssm(STRUCT_PFM, 0.0, fasta.STRUCT, None)}
    for thr in (float("-inf"), -3.0, 2.0):
        a = scanner.scan_combined(engine, recs, named, sp, tp, thr, "aligned", np.float64)
        b = scanner.scan_combined(OracleEngine(), recs, named, sp, tp, thr, "aligned", np.float64)
        assert len(a) == len(b)
        pd.testing.assert_frame_equal(a.drop(columns=["LogOdds.Struct", "LogOdds.SeqStruct"]),
                                      b.drop(columns=["LogOdds.Struct", "LogOdds.SeqStruct"]))
        for col in ("LogOdds.Struct", "LogOdds.SeqStruct"):
            x, y = a[col].to_numpy(), b[col].to_numpy()
            big = np.abs(y) > 1e9
            assert np.abs(x[~big] - y[~big]).max(initial=0) <= 1e-6 and np.allclose(x[big], y[big], rtol=1e-12)
        s1 = scanner.scan_records(engine, recs, sp, fasta.RNA, thr)
        s2 = scanner.scan_records(OracleEngine(), recs, sp, fasta.RNA, thr)
        pd.testing.assert_frame_equal(s1, s2)
