"""PFM files -> log-odds PSSM operands (host side, numpy only).

Mirrors ``pfm2pssm`` (rnascan/rnascan.py:238-252) without Biopython:

  * the PFM file is a TSV whose first column (``PO``) is dropped (:242-243);
  * ``Motif(counts=...).counts.normalize(pseudocount)``: add the scalar
    pseudocount to every cell, divide each position by its letter-sum;
  * ``.log_odds(background)``: background ``None`` -> uniform, otherwise it is
    renormalised to sum 1 over the alphabet; cell = log2(p / b), with
    p = 0 -> -inf, b = 0 < p -> +inf, both 0 -> NaN.

``normalize`` / ``log_odds`` live in Biopython (``biopython >= 1.66``,
setup.py:68), which is not part of the reference tree: the two functions
restate its documented behaviour, no reference fixture pins them ("parity
unpinned").
"""
import math
import os
from collections import OrderedDict

import numpy as np

from . import pack


def read_pfm(pfm_file):
    """PFM TSV -> OrderedDict letter -> float64 array (file column order).
    First column is the position index and is dropped (rnascan.py:242-243)."""
    with open(pfm_file) as fh:
        lines = [ln.rstrip("\n").rstrip("\r") for ln in fh if ln.strip()]
    if not lines:
        raise ValueError("empty PFM file %s" % pfm_file)
    header = lines[0].split("\t")
    letters = header[1:]
    cols = [[] for _ in letters]
    for ln in lines[1:]:
        parts = ln.split("\t")
        if len(parts) != len(header):
            raise ValueError("ragged PFM row in %s: %r" % (pfm_file, ln))
        for k in range(len(letters)):
            cols[k].append(float(parts[k + 1]))
    return OrderedDict((l, np.array(c, dtype=np.float64)) for l, c in zip(letters, cols))


def normalize(counts, pseudocount=0.0):
    letters = list(counts.keys())
    M = np.stack([np.asarray(counts[l], dtype=np.float64) for l in letters], axis=1) + float(pseudocount)
    total = np.zeros(M.shape[0], dtype=np.float64)
    for k in range(M.shape[1]):                 # same left-to-right sum as a Python loop over letters
        total = total + M[:, k]
    with np.errstate(divide="ignore", invalid="ignore"):
        M = M / total[:, None]
    return OrderedDict((l, M[:, k].copy()) for k, l in enumerate(letters))


def log_odds(pwm, background=None):
    letters = list(pwm.keys())
    if background is None:
        bg = {l: 1.0 / len(letters) for l in letters}
    else:
        total = 0.0
        for l in letters:
            total += float(background[l])
        bg = {l: float(background[l]) / total for l in letters}
    out = OrderedDict()
    for l in letters:
        b = bg[l]
        col = []
        for p in np.asarray(pwm[l], dtype=np.float64).tolist():
            # math.log(x, 2) (= log(x)/log(2)) as Biopython computes it, not log2:
            # the two differ in the last bit for some inputs
            if b > 0:
                q = p / b
                col.append(math.log(q, 2) if q > 0 else -math.inf)
            else:
                col.append(math.inf if p > 0 else math.nan)
        out[l] = np.array(col, dtype=np.float64)
    return out


class PSSM(OrderedDict):
    """letter -> float64[m] log-odds, plus the bits of Biopython's
    PositionSpecificScoringMatrix the scan path touches (``length``, ``alphabet``
    letters)."""

    def __init__(self, letters, values):
        OrderedDict.__init__(self)
        for l in letters:                       # filled in alphabet.letters order
            self[l] = np.asarray(values[l], dtype=np.float64)
        self.letters = "".join(letters)

    @property
    def length(self):
        return len(next(iter(self.values())))

    def letter_table(self, letters):
        """[m][8] table for the kernels: column c = letters[c], the rest NaN."""
        m = self.length
        T = np.full((m, pack.SEP + 1), np.nan, dtype=np.float64)
        for c, l in enumerate(letters):
            T[:, c] = self[l]
        return T

    def matrix(self, columns):
        """[m][len(columns)] matrix with the given letter per column."""
        return np.stack([self[l] for l in columns], axis=1)

    # -- the two Biopython / BioAddons methods the reference's scan path calls, on the GPU ---------------------
    def _is_rna(self):
        return set(self.letters) == set("GAUC")

    def calculate(self, sequence, engine=None):
        """ExtendedPositionSpecificScoringMatrix.calculate (matrix.py:68-81): every window score of ``sequence``.
        Nucleotide alphabets take the `_pwm.calculate` route (float32 ndarray, NaN for windows with a foreign letter,
        matrix.py:57-60); other alphabets the `_py_calculate` one (list of Python floats, matrix.py:25-43).  A single
        window gives a scalar (matrix.py:78-79)."""
        from . import compat
        engine = engine or compat.default_engine()
        sequence = str(sequence)
        m = self.length
        if self._is_rna():
            scores = engine.pwm_calculate(sequence, self.matrix(pack.RNA_LETTERS))
        else:
            stream = pack.pack([pack.encode_letters(sequence, self.letters)])
            full = engine.scan_letters_f64(stream, self.letter_table(self.letters))
            scores = [float(x) for x in full[:max(len(sequence) - m + 1, 0)]]
        if len(scores) == 1:
            return scores[0]
        return scores

    def search(self, sequence, threshold=0.0, both=False, engine=None):
        """PositionSpecificScoringMatrix.search as rnascan.py:263 uses it: yields ``(position, score)`` for every
        window with ``score > threshold`` (strict: NaN and -inf never pass), position 0-based, in window order; the
        sequence is upper-cased first.  One launch for the whole sequence (pfmscan_hits_host) instead of one
        `calculate` call per window.  ``both=True`` (reverse strand) is not part of the reference's call."""
        if both:
            raise ValueError("search(both=True) is not supported: rnascan only scans the given strand (rnascan.py:263)")
        from . import compat
        engine = engine or compat.default_engine()
        sequence = str(sequence).upper()
        m = self.length
        threshold = float(threshold)
        if self._is_rna():
            stream = pack.pack([pack.encode_rna(sequence)])
            table = self.letter_table(pack.RNA_LETTERS)
            if np.isneginf(threshold):
                sq, _ = engine.scan(stream, table, None)
                pos = np.flatnonzero(stream.window_mask(m) & (sq.astype(np.float64) > threshold))
                sc = sq[pos]
            else:
                pos, sc, _ = engine.hits(stream, table, None, threshold, -np.inf)
            for p, x in zip(pos.tolist(), sc):
                yield p, x                                  # numpy.float32, as calculate() returns it
        else:
            stream = pack.pack([pack.encode_letters(sequence, self.letters)])
            full = engine.scan_letters_f64(stream, self.letter_table(self.letters))
            keep = np.flatnonzero(stream.window_mask(m) & (full > threshold))
            for p in keep.tolist():
                yield p, float(full[p])


def pfm2pssm(pfm_file, pseudocount, letters, background=None):
    """rnascan.py:238-252.  ``letters`` = alphabet.letters (the order Biopython
    fills the matrix in); a PFM lacking one of them raises KeyError like the
    reference (load_motif prints "Check that you are using the correct --type")."""
    counts = read_pfm(pfm_file)
    counts = OrderedDict((l, counts[l]) for l in letters)       # KeyError on a wrong alphabet
    return PSSM(letters, log_odds(normalize(counts, pseudocount), background))


def is_multi_pfm(pfm_file):
    """multi-PFM files (pfmutil.py:115-133) start with a ``#id`` line"""
    with open(pfm_file) as fh:
        return fh.readline().startswith("#")


def read_multi_pfm(pfm_file):
    """The reference's multi-PFM format (reader pfmutil.py:89-113, writer :115-133):
    blocks of ``#<id>`` / ``#PO\t<letters>`` header lines followed by rows
    ``<pos>\t<v>...``, separated by blank lines.  Yields (id, OrderedDict letter -> array)."""
    with open(pfm_file) as fh:
        lines = [ln.rstrip("\r\n") for ln in fh]
    i, n = 0, len(lines)
    while i < n:
        if not lines[i].startswith("#"):
            i += 1
            continue
        header = []
        while i < n and lines[i].startswith("#"):
            header.append(lines[i][1:])
            i += 1
        if len(header) < 2:
            raise ValueError("multi-PFM block without a #PO header line in %s" % pfm_file)
        motif_id = header[0]
        letters = header[1].split("\t")[1:]
        cols = [[] for _ in letters]
        while i < n and not lines[i].startswith("#"):
            row = lines[i].rstrip()
            i += 1
            if not row:
                continue
            parts = row.split("\t")
            for k in range(len(letters)):
                cols[k].append(float(parts[k + 1]))
        yield motif_id, OrderedDict((l, np.array(c, dtype=np.float64)) for l, c in zip(letters, cols))


def load_pssms(pfm_file, pseudocount, letters, background=None):
    """OrderedDict motif id -> PSSM for a single-PFM file (one entry, id from the file
    name as rnascan.py:217) or a multi-PFM library (one entry per block)."""
    out = OrderedDict()
    if is_multi_pfm(pfm_file):
        for motif_id, counts in read_multi_pfm(pfm_file):
            counts = OrderedDict((l, counts[l]) for l in letters)
            out[motif_id] = PSSM(letters, log_odds(normalize(counts, pseudocount), background))
    else:
        out[motif_id_of(pfm_file)] = pfm2pssm(pfm_file, pseudocount, letters, background)
    return out


def motif_id_of(pfm_file):
    return os.path.splitext(os.path.basename(pfm_file))[0]      # rnascan.py:217
