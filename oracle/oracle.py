"""oracle.py -- TEST INFRASTRUCTURE ONLY.

Python face of the CPU oracle: ctypes bindings for ``liboracle.so`` (the plain-C
restatement in ``pfm_oracle.c``), a loader for ``_ref/_refpwm`` (the reference's
own ``_pwm.c`` compiled where it lies) and small pure-Python loop restatements
of the host arithmetic that lives in un-vendored Biopython.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; nothing under ``rnascan_amd/`` does.

Reference lines restated (relative to the upstream checkout, v0.10.2):
  * ``pwm_calculate``              <- rnascan/BioAddons/motifs/_pwm.c:34-68
  * ``py_calculate``               <- rnascan/BioAddons/motifs/matrix.py:25-43
  * ``scan_averaged_structure``    <- rnascan/rnascan.py:302-310
  * ``combine_keys``               <- rnascan/rnascan.py:422-433
  * ``compute_background``         <- rnascan/rnascan.py:444-457
  * ``normalize`` / ``log_odds``   <- Biopython (``biopython >= 1.66``, setup.py:68,
    NOT in the reference tree): documented behaviour of
    ``FrequencyPositionMatrix.normalize`` and ``PositionWeightMatrix.log_odds``.
    **parity unpinned** -- no reference test or fixture covers them.
"""
import ctypes
import glob
import importlib.util
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SEP = 7                      # separator / foreign code of the packed stream
RNA_LETTERS = "ACGU"         # sorted(alphabet.letters), matrix.py:57
STRUCT_LETTERS_SORTED = "BEHLMRT"   # file order written by pfmutil.py:62-70
STRUCT_LETTERS_ALPHABET = "EHTBLRM"  # BioAddons/Alphabet/__init__.py:24

_c_i64 = ctypes.c_int64
_p = ctypes.c_void_p


def build(force=False):
    """Compile liboracle.so (and _ref/ when the reference tree is present)."""
    lib = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pfm_oracle.c")
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not glob.glob(os.path.join(_HERE, "_ref", "_refpwm*.so"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        L = ctypes.CDLL(path)
        L.oracle_pwm_calculate.argtypes = [ctypes.c_char_p, _c_i64, _p, _c_i64, _p]
        L.oracle_py_calculate.argtypes = [ctypes.c_char_p, _c_i64, ctypes.c_char_p, ctypes.c_int, _p, _c_i64, _p]
        L.oracle_scan_averaged_structure.argtypes = [_p, _c_i64, ctypes.c_int, _p, _c_i64, _p]
        L.oracle_stream_seq.argtypes = [_p, _c_i64, _p, ctypes.c_int, _p]
        L.oracle_stream_letters_f64.argtypes = [_p, _c_i64, _p, ctypes.c_int, _p]
        L.oracle_stream_struct_f32.argtypes = [_p, _c_i64, _p, ctypes.c_int, _p]
        L.oracle_stream_struct_f64.argtypes = [_p, _c_i64, _p, ctypes.c_int, _p]
        L.oracle_stream_seqstruct_f32.argtypes = [_p, _p, _c_i64, _p, _p, ctypes.c_int, _p, _p]
        L.oracle_stream_hits.argtypes = [_p, _p, _c_i64, ctypes.c_double, ctypes.c_double, _c_i64, _p]
        L.oracle_stream_hits.restype = _c_i64
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(_p) if a is not None else None


def ref_pwm():
    """The reference's own _pwm.c loop (oracle/_ref), or None when not built."""
    hits = glob.glob(os.path.join(_HERE, "_ref", "_refpwm*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("_refpwm", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# --------------------------------------------------------------------------
# window scoring
# --------------------------------------------------------------------------
def pwm_calculate(sequence, matrix):
    """_pwm.c:34-68 -- str + float64[m][4] (A,C,G,U columns) -> float32[n]."""
    seq = sequence.encode("ascii") if isinstance(sequence, str) else bytes(sequence)
    M = np.ascontiguousarray(matrix, dtype=np.float64)
    assert M.ndim == 2 and M.shape[1] == 4
    m = M.shape[0]
    n = max(len(seq) - m + 1, 0)
    out = np.empty(n, dtype=np.float32)
    if n:
        lib().oracle_pwm_calculate(seq, len(seq), _ptr(M), m, _ptr(out))
    return out


def py_calculate(sequence, letters, table):
    """matrix.py:25-43 -- generic alphabet, fp64 list, NaN+break on KeyError."""
    seq = sequence.encode("ascii")
    T = np.ascontiguousarray(table, dtype=np.float64)
    m, nl = T.shape
    assert nl == len(letters)
    n = max(len(seq) - m + 1, 0)
    out = np.empty(n, dtype=np.float64)
    if n:
        lib().oracle_py_calculate(seq, len(seq), letters.encode("ascii"), nl, _ptr(T), m, _ptr(out))
    return out


def scan_averaged_structure(profile, pssm):
    """rnascan.py:302-307 -- all window scores (fp64); columns already paired."""
    prof = np.ascontiguousarray(profile, dtype=np.float64)
    P = np.ascontiguousarray(pssm, dtype=np.float64)
    L, nc = prof.shape
    N = P.shape[0]
    assert P.shape[1] == nc
    n = max(L - N + 1, 0)
    out = np.empty(n, dtype=np.float64)
    if n:
        lib().oracle_scan_averaged_structure(_ptr(prof), L, nc, _ptr(P), N, _ptr(out))
    return out


def stream_seq(codes, table8):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    T = np.ascontiguousarray(table8, dtype=np.float64)
    assert T.shape[1] == 8
    out = np.empty(codes.size, dtype=np.float32)
    lib().oracle_stream_seq(_ptr(codes), codes.size, _ptr(T), T.shape[0], _ptr(out))
    return out


def stream_letters_f64(codes, table8):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    T = np.ascontiguousarray(table8, dtype=np.float64)
    assert T.shape[1] == 8
    out = np.empty(codes.size, dtype=np.float64)
    lib().oracle_stream_letters_f64(_ptr(codes), codes.size, _ptr(T), T.shape[0], _ptr(out))
    return out


def stream_struct(profile, pssm7):
    P = np.ascontiguousarray(pssm7, dtype=np.float64)
    assert P.shape[1] == 7 and profile.shape[1] == 7
    out = np.empty(profile.shape[0], dtype=np.float64)
    if profile.dtype == np.float32:
        prof = np.ascontiguousarray(profile)
        lib().oracle_stream_struct_f32(_ptr(prof), prof.shape[0], _ptr(P), P.shape[0], _ptr(out))
    else:
        prof = np.ascontiguousarray(profile, dtype=np.float64)
        lib().oracle_stream_struct_f64(_ptr(prof), prof.shape[0], _ptr(P), P.shape[0], _ptr(out))
    return out


def stream_hits(seq, st, thr_seq, thr_struct):
    """positions p with seq[p] > thr_seq and st[p] > thr_struct (either may be None)."""
    n = (seq if seq is not None else st).shape[0]
    pos = np.empty(n, dtype=np.int64)
    k = lib().oracle_stream_hits(_ptr(seq), _ptr(st), n, float(thr_seq), float(thr_struct), n, _ptr(pos))
    return pos[:k].copy()


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads.argtypes = [ctypes.c_int]
    lib().oracle_set_num_threads(int(n))


# --------------------------------------------------------------------------
# host arithmetic restated with plain loops
# --------------------------------------------------------------------------
def normalize(counts, pseudocount=0.0):
    """Biopython FrequencyPositionMatrix.normalize (documented behaviour; parity
    unpinned): add the scalar pseudocount to every cell, divide each position by
    its letter-sum.  ``counts``: dict letter -> list."""
    letters = list(counts.keys())
    length = len(counts[letters[0]])
    out = {l: [0.0] * length for l in letters}
    for i in range(length):
        total = 0.0
        for l in letters:
            total += counts[l][i] + pseudocount
        for l in letters:
            out[l][i] = (counts[l][i] + pseudocount) / total
    return out


def log_odds(pwm, background=None):
    """Biopython PositionWeightMatrix.log_odds (documented behaviour; parity
    unpinned): background None -> uniform, else renormalised to sum 1;
    cell = log2(p/b); p=0,b>0 -> -inf; p>0,b=0 -> +inf; both 0 -> NaN."""
    letters = list(pwm.keys())
    if background is None:
        bg = {l: 1.0 / len(letters) for l in letters}
    else:
        total = sum(float(background[l]) for l in letters)
        bg = {l: float(background[l]) / total for l in letters}
    out = {}
    for l in letters:
        b = bg[l]
        row = []
        for p in pwm[l]:
            if b > 0:
                p = p / b
                if p > 0:
                    v = math.log(p, 2)
                else:
                    v = -math.inf
            else:
                if p > 0:
                    v = math.inf
                else:
                    v = math.nan
            row.append(v)
        out[l] = row
    return out


def compute_background(sequences, letters):
    """rnascan.py:444-457: counts per alphabet letter over all (already
    preprocessed) sequences, +1 pseudocount per letter, divided by total+len."""
    content = {l: 0 for l in letters}
    total = len(letters)
    for s in sequences:
        for l in letters:
            c = s.count(l)
            content[l] += c
            total += c
    return {l: (float(c) + 1) / total for l, c in content.items()}


def combine_keys(seq_hits, struct_hits):
    """rnascan.py:422-423: inner join on (Sequence_ID, Start, End).  Inputs are
    lists of (seq_id, start, end, score); returns list of
    (seq_id, start, end, seq_score, struct_score, sum) in seq-table order."""
    index = {}
    for sid, st, en, sc in struct_hits:
        index.setdefault((sid, st, en), []).append(sc)
    out = []
    for sid, st, en, sc in seq_hits:
        for sc2 in index.get((sid, st, en), []):
            out.append((sid, st, en, sc, sc2, float(sc) + sc2))
    return out
