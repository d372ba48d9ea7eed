#!/usr/bin/env python3
"""Regenerate tests/golden/ -- runs ONLY in the build container, where the
upstream checkout is mounted read-only at /root/reference.  The GPU box never
sees the reference; it gets the small data fixtures this script writes.

What is captured, and from what:

  (1) ``pwm`` / ``calculate_route``: the reference's C loop
      (rnascan/BioAddons/motifs/_pwm.c:7-70) compiled from where it lies by
      oracle/Makefile into oracle/_ref/_refpwm*.so, called directly and through
      the reference's own ``ExtendedPositionSpecificScoringMatrix.calculate``
      (rnascan/BioAddons/motifs/matrix.py:50-81).
  (2) ``py_calculate``: matrix.py:25-43, run unmodified.
  (3) ``scan_averaged_structure`` / ``scan_main_dir`` / ``combine``:
      rnascan/rnascan.py:293-315, :335-413 (directory branch), :416-434, run
      unmodified.
  (4) ``ref_tests``: the known answers the reference's own tests hold
      (tests/motif_scan_test.py:37-40, tests/preprocess_seq_test.py) -- data only.
  (5) ``data/``: the reference's example/ and tests/ DATA files (PFM tables, the
      FASTA records, one averaged-structure profile).  No source text.

How the Python functions are imported: Biopython is not installed in this image
and is not in the reference tree, so ``import rnascan.rnascan`` fails on its
module-level ``from Bio import ...`` lines.  The names those lines need are
registered in ``sys.modules`` as EMPTY placeholders (class names only, plus a
plain ``dict`` subclass for ``PositionSpecificScoringMatrix`` that stores the
letter->list mapping it is given).  No Biopython ARITHMETIC is stood in for:
``normalize``, ``log_odds``, ``search``, ``consensus``, ``transcribe``, ``count``
and ``SeqIO`` are absent and every reference function that needs them
(``scan``, ``scan_all``, ``pfm2pssm``, ``preprocess_seq``, ``compute_background``,
``parse_sequences``) is left un-run -- their parity stays "unpinned"
(DESIGN.md).  The PSSM operands fed to the reference functions are built by
oracle.normalize/log_odds and recorded in the fixture next to the outputs, so
each vector pins "reference(scan) given this PSSM".

Run:  python3 -B tests/golden/make_golden.py
"""
import argparse
import json
import os
import shutil
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
from oracle import oracle  # noqa: E402

DATA_FILES = [
    "example/HIST2H3C_3p_end.fa",
    "example/HIST2H3C_3p_end_structure.txt",
    "example/SLBP_pfm_assembled_normalized_seq.txt",
    "example/SLBP_pfm_assembled_normalized_struct.txt",
    "example/3p_UTR_background_structural_context.txt",
    "tests/test.fa",
    "tests/test_seq_pfm.txt",
    "tests/test_struct_pfm.txt",
]


# ---------------------------------------------------------------------------
# placeholder modules so the reference's module-level imports resolve
# ---------------------------------------------------------------------------
class Alphabet(object):
    letters = None


class SingleLetterAlphabet(Alphabet):
    pass


class NucleotideAlphabet(SingleLetterAlphabet):
    pass


class RNAAlphabet(NucleotideAlphabet):
    pass


class SecondaryStructure(SingleLetterAlphabet):
    pass


class IUPACAmbiguousRNA(RNAAlphabet):
    letters = "GAUCRYWSMKHBVDN"


class IUPACUnambiguousRNA(IUPACAmbiguousRNA):
    letters = "GAUC"


class SortedStruct(SecondaryStructure):
    letters = "BEHLMRT"     # same order as the averaged-structure file's header


class PositionSpecificScoringMatrix(dict):
    """letter -> list container; fills in alphabet.letters order."""

    def __init__(self, alphabet, values):
        dict.__init__(self)
        self.alphabet = alphabet
        self.length = None
        for letter in alphabet.letters:
            self[letter] = list(values[letter])
            self.length = len(self[letter])

    def __reduce__(self):        # Pool.map pickles the PSSM (rnascan.py:363-366)
        return (_rebuild, (type(self), self.alphabet, dict(self)))


def _rebuild(cls, alphabet, values):
    return cls(alphabet, values)


class Seq(object):
    pass


class SeqRecord(object):
    pass


def install_placeholders(refpwm):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    bio = mod("Bio")
    iupac = mod("Bio.Alphabet.IUPAC", IUPACAmbiguousRNA=IUPACAmbiguousRNA,
                IUPACUnambiguousRNA=IUPACUnambiguousRNA)
    alph = mod("Bio.Alphabet", Alphabet=Alphabet, SingleLetterAlphabet=SingleLetterAlphabet,
               NucleotideAlphabet=NucleotideAlphabet, RNAAlphabet=RNAAlphabet,
               SecondaryStructure=SecondaryStructure, IUPAC=iupac)
    mat = mod("Bio.motifs.matrix", PositionSpecificScoringMatrix=PositionSpecificScoringMatrix)
    motifs = mod("Bio.motifs", matrix=mat)
    seqio = mod("Bio.SeqIO")
    seq = mod("Bio.Seq", Seq=Seq)
    seqrec = mod("Bio.SeqRecord", SeqRecord=SeqRecord)
    bio.Alphabet, bio.motifs, bio.SeqIO, bio.Seq, bio.SeqRecord = alph, motifs, seqio, seq, seqrec
    # the compiled reference loop under the name matrix.py:47 imports
    import rnascan.BioAddons.motifs as ref_motifs_pkg
    sys.modules["rnascan.BioAddons.motifs._pwm"] = refpwm
    ref_motifs_pkg._pwm = refpwm
    return types.SimpleNamespace(IUPACUnambiguousRNA=IUPACUnambiguousRNA,
                                 SecondaryStructure=SecondaryStructure,
                                 PSSM=PositionSpecificScoringMatrix)


def read_pfm_table(path):
    """PFM TSV -> (letters in file order, dict letter -> list)."""
    df = pd.read_csv(path, sep="\t")
    letters = list(df.columns[1:])
    return letters, {l: [float(x) for x in df[l]] for l in letters}


def build_pssm(path, pseudocount, background, alphabet_letters):
    """counts are taken in alphabet.letters order, the order Biopython's Motif stores
    them in and sums them in when it normalises (GAUC / EHTBLRM), not file order"""
    letters, counts = read_pfm_table(path)
    counts = {l: counts[l] for l in alphabet_letters}
    return letters, oracle.log_odds(oracle.normalize(counts, pseudocount), background)


def fasta_records(path):
    recs, name, buf = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(buf)))
            name, buf = line[1:], []
        else:
            buf.append(line.strip())
    if name is not None:
        recs.append((name, "".join(buf)))
    return recs


def f32list(a):
    return [float(x) for x in np.asarray(a, dtype=np.float32)]


def f64list(a):
    return [float(x) for x in np.asarray(a, dtype=np.float64)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    out_dir = args.out
    data_dir = os.path.join(out_dir, "data")
    os.makedirs(data_dir, exist_ok=True)

    oracle.build()
    refpwm = oracle.ref_pwm()
    assert refpwm is not None, "oracle/_ref not built (needs /root/reference)"
    sys.path.insert(0, REF)
    ns = install_placeholders(refpwm)
    import rnascan.rnascan as ms                      # the reference, unmodified
    from rnascan.BioAddons.motifs.matrix import ExtendedPositionSpecificScoringMatrix as RefPSSM
    from rnascan.BioAddons.Alphabet import ContextualSecondaryStructure

    # ---- (5) data files -----------------------------------------------------
    for rel in DATA_FILES:
        shutil.copyfile(os.path.join(REF, rel), os.path.join(data_dir, os.path.basename(rel)))

    G = {"_about": "generated by tests/golden/make_golden.py; see its docstring"}

    # ---- (4) known answers held by the reference's own tests -----------------
    G["ref_tests"] = {
        "compute_background": {          # tests/motif_scan_test.py:31-43, assertAlmostEqual(..., 3)
            "fasta": "test.fa", "places": 3,
            "expected": {"A": 0.1944, "C": 0.1388, "U": 0.5277, "G": 0.1388},
        },
        "preprocess_seq": [              # tests/preprocess_seq_test.py:13-53
            {"seq": "GATTACA", "source": "DNA", "target": "RNA", "expected": "GAUUACA"},
            {"seq": "GAUUACA", "source": "RNA", "target": "RNA", "expected": "GAUUACA"},
            {"seq": "GAUUACA", "source": "RNA", "target": "DNA", "expected": "GAUUACA"},
            {"seq": "GATTACA", "source": "DNA", "target": "DNA", "expected": "GATTACA"},
            {"seq": "GAUUACA", "source": "generic", "target": "RNA", "expected": "GAUUACA"},
            {"seq": "KHIL", "source": "struct", "target": "RNA", "expected": "KHIL"},
        ],
    }

    rng = np.random.default_rng(20240601)
    hist_name, hist_dna = fasta_records(os.path.join(REF, "example/HIST2H3C_3p_end.fa"))[0]
    hist_rna = hist_dna.replace("T", "U").replace("t", "u").upper()   # rnascan.py:191-193
    slbp_seq = os.path.join(REF, "example/SLBP_pfm_assembled_normalized_seq.txt")
    slbp_struct = os.path.join(REF, "example/SLBP_pfm_assembled_normalized_struct.txt")
    hist_profile = os.path.join(REF, "example/HIST2H3C_3p_end_structure.txt")

    # ---- (1) reference C loop -------------------------------------------------
    def matrix_acgu(pssm):
        m = len(pssm["A"])
        return np.array([[pssm[l][i] for l in "ACGU"] for i in range(m)], dtype=np.float64)

    pwm_cases = []

    def add_pwm(name, sequence, M, note=""):
        scores = refpwm.calculate(sequence, M)
        assert scores.dtype == np.float32
        pwm_cases.append({"name": name, "note": note, "sequence": sequence,
                          "matrix": [f64list(r) for r in M], "scores": f32list(scores)})

    _, pssm0 = build_pssm(slbp_seq, 0.0, None, "GAUC")
    add_pwm("hist_slbp_pc0_uniform", hist_rna, matrix_acgu(pssm0), "config 1 seq side: -u -C 0")
    bg_hist = oracle.compute_background([hist_rna], "GAUC")
    _, pssm1 = build_pssm(slbp_seq, 0.01, bg_hist, "GAUC")
    add_pwm("hist_slbp_pc001_bg", hist_rna, matrix_acgu(pssm1), "computed background, -C 0.01")
    add_pwm("hist_dna_letters_mixed_case", hist_dna[:120] + hist_dna[120:].lower(), matrix_acgu(pssm0),
            "T/t accepted like U/u, lower case accepted (_pwm.c:47-60)")
    _, pssm_t = build_pssm(os.path.join(REF, "tests/test_seq_pfm.txt"), 0.0, None, "GAUC")
    for rid, s in fasta_records(os.path.join(REF, "tests/test.fa")):
        add_pwm("testfa_%s_test_seq_pfm" % rid, s, matrix_acgu(pssm_t), "tests/ PFM has header order G A U C")
    letters_pool = np.array(list("ACGU"))
    for m in (1, 2, 8, 12, 18, 33):
        seq = "".join(rng.choice(letters_pool, size=400))
        seq = list(seq)
        for pos in rng.choice(400, size=6, replace=False):
            seq[pos] = "N"
        for pos in rng.choice(400, size=40, replace=False):
            seq[pos] = seq[pos].lower()
        seq = "".join(seq)
        M = rng.normal(0, 2, size=(m, 4))
        add_pwm("synthetic_m%d" % m, seq, M, "random PSSM; N poisons covering windows; lower case")
    Minf = rng.normal(0, 2, size=(8, 4))
    Minf[2, 1] = -np.inf
    Minf[5, 3] = -np.inf
    Minf[6, 0] = np.inf
    add_pwm("synthetic_inf_cells", "".join(rng.choice(letters_pool, size=300)), Minf,
            "-inf/+inf cells propagate; +inf-inf gives NaN")
    add_pwm("exact_one_window", "ACGUACGU", rng.normal(0, 2, size=(8, 4)), "len == m -> 1 score")
    add_pwm("one_shorter_than_motif", "ACGUACG", rng.normal(0, 2, size=(8, 4)), "len == m-1 -> 0 scores")
    try:
        refpwm.calculate("ACG", rng.normal(0, 2, size=(8, 4)))
        short_err = None
    except Exception as e:          # negative shape -> _pwm.c:26-31 raises MemoryError
        short_err = type(e).__name__
    G["pwm_shorter_than_m_minus_1_raises"] = short_err
    add_pwm("other_letters", "ACGU-RYKM*acguXACGUACGUACGU", rng.normal(0, 2, size=(4, 4)),
            "IUPAC ambiguity codes and punctuation are foreign")
    # a cancellation-heavy case where fp64->fp32 rounding depends on summation order
    Mc = rng.normal(0, 30, size=(12, 4))
    add_pwm("synthetic_large_magnitude", "".join(rng.choice(letters_pool, size=2000)), Mc,
            "sequential fp64 sum then one f32 cast")
    G["pwm"] = pwm_cases

    # ---- calculate() through matrix.py:50-81 ----------------------------------
    rna = ns.IUPACUnambiguousRNA()
    ref_pssm0 = RefPSSM(rna, ns.PSSM(rna, pssm0))
    route = ref_pssm0.calculate(hist_rna)
    one = ref_pssm0.calculate(hist_rna[:18])
    G["calculate_route"] = {
        "note": "ExtendedPositionSpecificScoringMatrix.calculate -> _calculate -> _pwm.calculate; letters sorted -> ACGU; one window -> scalar",
        "pssm_letters_order": list(rna.letters), "pssm": {l: f64list(pssm0[l]) for l in rna.letters},
        "sequence": hist_rna, "scores": f32list(route),
        "single_window_sequence": hist_rna[:18], "single_window_is_scalar": bool(np.ndim(one) == 0),
        "single_window_score": float(one),
    }

    # ---- (2) _py_calculate -----------------------------------------------------
    css = ContextualSecondaryStructure()
    _, spssm0 = build_pssm(slbp_struct, 0.0, None, "EHTBLRM")
    _, spssm1 = build_pssm(slbp_struct, 0.01, None, "EHTBLRM")
    py_cases = []

    def add_py(name, sequence, pssm, m_override=None):
        ref = RefPSSM(css, ns.PSSM(css, pssm))
        m = ref.length if m_override is None else m_override
        scores = ref._py_calculate(sequence, m, len(sequence))
        py_cases.append({"name": name, "sequence": sequence, "letters": css.letters, "m": m,
                         "table": [[pssm[l][i] for l in css.letters] for i in range(ref.length)],
                         "scores": [float(x) for x in scores]})

    add_py("EELLX_m2", "EELLX", spssm0, 2)
    struct_pool = np.array(list("EHTBLRM"))
    s1 = "".join(rng.choice(struct_pool, size=120))
    add_py("random_struct_pc0", s1, spssm0)
    add_py("random_struct_pc001_lower", s1[:60].lower() + s1[60:], spssm1)
    s2 = list("".join(rng.choice(struct_pool, size=90)))
    s2[17] = "X"
    s2[60] = "."
    add_py("random_struct_foreign", "".join(s2), spssm1)
    G["py_calculate"] = py_cases

    # ---- (3) scan_averaged_structure -------------------------------------------
    avg_cases = []

    def run_avg(name, profile_path, pssm, pairing, minscore, store_profile=False):
        alpha = css if pairing == "positional" else SortedStruct()
        ref = RefPSSM(alpha, ns.PSSM(alpha, pssm))
        df = ms.scan_averaged_structure(profile_path, {"motif": ref}, minscore)
        rows = [] if df.shape[0] == 0 else [[int(r.Start), int(r.End), float(r.LogOdds)] for r in df.itertuples()]
        case = {"name": name, "pairing": pairing, "minscore": minscore,
                "pssm_letter_order": alpha.letters,
                "pssm": [[pssm[l][i] for l in alpha.letters] for i in range(len(pssm["E"]))],
                "profile_file": os.path.basename(profile_path) if not store_profile else None,
                "columns": [] if df.shape[0] == 0 else list(df.columns),
                "sequence_field": None if df.shape[0] == 0 else str(df.iloc[0]["Sequence"]),
                "rows": rows}
        if store_profile:
            prof = pd.read_csv(profile_path, sep="\t")
            case["profile_header"] = list(prof.columns)
            case["profile"] = [[float(x) for x in row[1:]] for row in prof.itertuples(index=False)]
        avg_cases.append(case)

    ninf = float("-inf")
    run_avg("hist_slbp_pc0_positional", hist_profile, spssm0, "positional", ninf)
    run_avg("hist_slbp_pc0_aligned", hist_profile, spssm0, "aligned", ninf)
    run_avg("hist_slbp_pc001_aligned_thr0", hist_profile, spssm1, "aligned", 0.0)
    run_avg("hist_slbp_pc001_positional_thr_m20", hist_profile, spssm1, "positional", -20.0)
    # synthetic profile: Dirichlet(0.3) rows with small entries snapped to exact 0
    tmp = os.path.join("/tmp", "golden_profiles")
    os.makedirs(tmp, exist_ok=True)
    prof = rng.dirichlet(np.full(7, 0.3), size=60)
    prof[prof < 0.02] = 0.0
    prof /= prof.sum(axis=1, keepdims=True)
    synth_path = os.path.join(tmp, "structure.synth1.txt")
    with open(synth_path, "w") as f:
        f.write("PO\t" + "\t".join("BEHLMRT") + "\n")
        for i, row in enumerate(prof):
            f.write(str(i) + "\t" + "\t".join(str(float(x)) for x in row) + "\n")
    synth_counts = {l: list(rng.dirichlet(np.full(7, 0.5), size=12)[:, k]) for k, l in enumerate("EHTBLRM")}
    zero_mask = rng.random((12, 7)) < 0.15
    for k, l in enumerate("EHTBLRM"):
        for i in range(12):
            if zero_mask[i, k]:
                synth_counts[l][i] = 0.0
    synth_pssm0 = oracle.log_odds(oracle.normalize(synth_counts, 0.0), None)
    synth_pssm1 = oracle.log_odds(oracle.normalize(synth_counts, 0.01), None)
    run_avg("synth_w12_pc0_aligned", synth_path, synth_pssm0, "aligned", ninf, store_profile=True)
    run_avg("synth_w12_pc001_aligned", synth_path, synth_pssm1, "aligned", ninf, store_profile=True)
    run_avg("synth_w12_pc0_positional", synth_path, synth_pssm0, "positional", ninf, store_profile=True)
    # a longer profile, width 18, background with a ZERO entry -> +inf log-odds next to the -inf ones:
    # +DBL_MAX / -DBL_MAX row-dots, inf - inf -> NaN -> 0 inside one window (rnascan.py:306)
    prof2 = rng.dirichlet(np.full(7, 0.25), size=150)
    prof2[prof2 < 0.03] = 0.0
    prof2 /= prof2.sum(axis=1, keepdims=True)
    synth2_path = os.path.join(tmp, "structure.synth2.txt")
    with open(synth2_path, "w") as f:
        f.write("PO\t" + "\t".join("BEHLMRT") + "\n")
        for i, row in enumerate(prof2):
            f.write(str(i) + "\t" + "\t".join(str(float(x)) for x in row) + "\n")
    counts18 = {l: list(rng.dirichlet(np.full(7, 0.5), size=18)[:, k]) for k, l in enumerate("EHTBLRM")}
    zm = rng.random((18, 7)) < 0.12
    for k, l in enumerate("EHTBLRM"):
        for i in range(18):
            if zm[i, k]:
                counts18[l][i] = 0.0
    for i in range(18):                       # letter E: background 0; counts 0 except two rows ->
        if i not in (3, 11):                  # NaN cells (0/0) almost everywhere, +inf in rows 3 and 11
            counts18["E"][i] = 0.0
    bg_zero = {"E": 0.0, "H": 0.2, "T": 0.1, "B": 0.1, "L": 0.3, "R": 0.2, "M": 0.1}
    pssm18 = oracle.log_odds(oracle.normalize(counts18, 0.0), bg_zero)
    run_avg("synth_w18_pc0_bgzero_aligned", synth2_path, pssm18, "aligned", ninf, store_profile=True)
    run_avg("synth_w18_pc0_bgzero_positional_thr", synth2_path, pssm18, "positional", -1e300, store_profile=True)
    counts8 = {l: list(rng.dirichlet(np.full(7, 0.5), size=8)[:, k]) for k, l in enumerate("EHTBLRM")}
    pssm8 = oracle.log_odds(oracle.normalize(counts8, 0.01), None)
    run_avg("synth_w8_pc001_positional", synth2_path, pssm8, "positional", -5.0, store_profile=True)
    G["scan_averaged_structure"] = avg_cases

    # ---- scan_main directory branch + combine + _add_match_id -------------------
    sdir = os.path.join(tmp, "avgdir")
    shutil.rmtree(sdir, ignore_errors=True)
    os.makedirs(sdir)
    shutil.copyfile(hist_profile, os.path.join(sdir, "structure.hg19_dna.txt"))
    aligned = SortedStruct()
    ref_struct = RefPSSM(aligned, ns.PSSM(aligned, spssm1))
    margs = argparse.Namespace(minscore=0.0, debug=False, cores=2)
    struct_df = ms.scan_main(sdir, {"SLBP_struct": ref_struct}, aligned, None, margs)
    G["scan_main_dir"] = {
        "note": "rnascan.py:348-375 pool form: Sequence_ID from ^structure\\.(.*)\\.txt$",
        "columns": list(struct_df.columns),
        "rows": [[str(r[0]), str(r[1]), str(r[2]), int(r[3]), int(r[4]), str(r[5]), float(r[6])]
                 for r in struct_df.itertuples(index=False)],
    }
    # seq table built the way scan()/scan_all()/scan_main produce it (rnascan.py:264-286, :401-413)
    # from the reference C loop's scores; threshold strict >, LogOdds = round(float32, 3)
    scores = refpwm.calculate(hist_rna, matrix_acgu(pssm1))
    rows = []
    for pos, sc in enumerate(scores):
        if sc > 0.0:
            rows.append(["SLBP_seq", pos + 1, pos + 18, hist_rna[pos:pos + 18], round(sc, 3)])
    seq_df = pd.DataFrame(rows, columns=["Motif_ID", "Start", "End", "Sequence", "LogOdds"])
    seq_df = seq_df.sort_values(["Start", "Motif_ID"])
    seq_df["Sequence_ID"] = "hg19_dna"
    seq_df["Description"] = hist_name
    cols = seq_df.columns.tolist()
    seq_df = seq_df[cols[-2:] + cols[:-2]]
    comb = ms.combine(seq_df, struct_df)
    ms._add_match_id(comb)
    G["combine"] = {
        "note": "rnascan.py:416-434 + :329-332 on the two tables above (minscore 0.0)",
        "seq_rows": [[str(r[0]), str(r[1]), str(r[2]), int(r[3]), int(r[4]), str(r[5]), float(r[6])]
                     for r in seq_df.itertuples(index=False)],
        "seq_logodds_dtype": str(seq_df["LogOdds"].dtype),
        "columns": list(comb.columns),
        "dtypes": {c: str(comb[c].dtype) for c in comb.columns},
        "rows": [[(x.item() if hasattr(x, "item") else x) for x in r] for r in comb.itertuples(index=False)],
        "tsv": comb.to_csv(sep="\t", index=False),
    }

    with open(os.path.join(out_dir, "golden.json"), "w") as f:
        json.dump(G, f, indent=1)
    print("wrote", os.path.join(out_dir, "golden.json"),
          "pwm cases:", len(pwm_cases), "py cases:", len(py_cases), "avg cases:", len(avg_cases))


if __name__ == "__main__":
    main()
