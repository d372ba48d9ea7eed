"""Packing of records into the stream layout of include/pfmscan.h.

Every record is followed by ONE separator position (code 7); windows that touch
it score NaN on the letter path, so no window can span two records and the
kernels need no per-record metadata.  Offsets are kept on the host to turn a
stream position back into (record, 0-based start).
"""
import numpy as np

SEP = 7
RNA_LETTERS = "ACGU"                 # sorted(alphabet.letters), matrix.py:57; column order of _pwm.c:45-60
STRUCT_LETTERS = "EHTBLRM"           # ContextualSecondaryStructure.letters, BioAddons/Alphabet/__init__.py:24
STRUCT_COLUMNS = "BEHLMRT"           # column order of averaged-structure files (pfmutil.py:62-70 sorts the keys)

_RNA_LUT = np.full(256, SEP, dtype=np.uint8)
for _i, _pair in enumerate(("Aa", "Cc", "Gg", "TtUu")):   # the switch of _pwm.c:41-63
    for _ch in _pair:
        _RNA_LUT[ord(_ch)] = _i


def encode_rna(seq):
    """ASCII nucleotide string -> codes, exactly the letter classes of _pwm.c:41-63
    (case-insensitive, T == U, everything else foreign)."""
    b = np.frombuffer(seq.encode("latin-1") if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    return _RNA_LUT[b]


CASE_BIT = 8                         # bit 3 of a generic-alphabet code: the input wrote the letter in lower case


def letter_lut(letters, keep_case=False):
    """LUT for a generic alphabet; ``_py_calculate`` upper-cases first (matrix.py:31), so both cases score alike.
    ``keep_case``: a lower-case letter gets its index | CASE_BIT -- the kernels read bits 0..2 only (include/pfmscan.h),
    and the ``Sequence`` column can show the structure string as it was written (rnascan.py:186-197 does not upper-case
    it, :272 slices the record itself)."""
    if len(letters) > SEP:
        raise ValueError("at most 7 letters per alphabet")
    lut = np.full(256, SEP, dtype=np.uint8)
    for i, ch in enumerate(letters):
        lut[ord(ch.upper())] = i
        if ch.lower() != ch.upper():
            lut[ord(ch.lower())] = i | (CASE_BIT if keep_case else 0)
    return lut


def encode_letters(seq, letters, keep_case=False):
    b = np.frombuffer(seq.encode("latin-1", "replace") if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    return letter_lut(letters, keep_case)[b]


class Stream(object):
    """A packed batch of records.

    codes    uint8 [n_pos] or None
    profile  float32/float64 [n_pos][7] or None
    offsets  int64 [R]   stream position of each record's first letter
    lengths  int64 [R]   record lengths (without the separator)
    """

    def __init__(self, codes, profile, offsets, lengths, codes2=None):
        self.codes = codes
        self.codes2 = codes2            # second code stream of the same records (two-FASTA combined scan), or None
        self.profile = profile
        self.offsets = offsets
        self.lengths = lengths
        self.n_pos = int(offsets[-1] + lengths[-1] + 1) if len(offsets) else 0

    @property
    def n_records(self):
        return len(self.offsets)

    def n_windows(self, m):
        return int(np.maximum(self.lengths - m + 1, 0).sum())

    def locate(self, pos):
        """stream positions -> (record index, 0-based start within the record)."""
        pos = np.asarray(pos, dtype=np.int64)
        if pos.size > 4 * self.offsets.size and bool(np.all(pos[1:] >= pos[:-1])):
            # many sorted positions (hit lists are): count the positions of every record instead of searching per position
            cuts = np.searchsorted(pos, self.offsets, side="left")
            rec = np.repeat(np.arange(self.offsets.size, dtype=np.int64), np.diff(np.append(cuts, pos.size)))
            if rec.size != pos.size:                       # positions before the first record: not a hit list
                rec = np.searchsorted(self.offsets, pos, side="right") - 1
        else:
            rec = np.searchsorted(self.offsets, pos, side="right") - 1
        return rec, pos - self.offsets[rec]

    def record_slice(self, r, m):
        """slice of a position-aligned score array holding record r's windows."""
        n = max(int(self.lengths[r]) - m + 1, 0)
        return slice(int(self.offsets[r]), int(self.offsets[r]) + n)

    def window_mask(self, m):
        """bool [n_pos]: True where a window of width m lies inside one record."""
        n = np.maximum(self.lengths - m + 1, 0)
        edge = np.zeros(self.n_pos + 1, dtype=np.int8)       # +1 at a record's first window, -1 after its last one
        np.add.at(edge, self.offsets, 1)
        np.add.at(edge, self.offsets + n, -1)
        return np.cumsum(edge[:-1], dtype=np.int8).astype(bool)


def pack(code_arrays=None, profiles=None, profile_dtype=np.float32):
    """Pack per-record code arrays and/or per-record [L][7] profiles into a Stream."""
    src = code_arrays if code_arrays is not None else profiles
    if src is None:
        raise ValueError("nothing to pack")
    lengths = np.array([len(x) for x in src], dtype=np.int64)
    if code_arrays is not None and profiles is not None:
        plen = np.array([len(x) for x in profiles], dtype=np.int64)
        if plen.shape != lengths.shape or np.any(plen != lengths):
            raise ValueError("sequence and profile lengths differ")
    offsets = np.zeros(len(lengths), dtype=np.int64)
    if len(lengths) > 1:
        offsets[1:] = np.cumsum(lengths[:-1] + 1)
    n_pos = int((lengths + 1).sum())
    codes = prof = None
    if code_arrays is not None:
        codes = np.full(n_pos, SEP, dtype=np.uint8)
        for off, c in zip(offsets, code_arrays):
            codes[off:off + len(c)] = c
    if profiles is not None:
        prof = np.zeros((n_pos, 7), dtype=profile_dtype)
        for off, p in zip(offsets, profiles):
            p = np.asarray(p)
            if p.ndim != 2 or p.shape[1] != 7:
                raise ValueError("profile must be [L][7]")
            prof[off:off + len(p)] = p
    return Stream(codes, prof, offsets, lengths)
