"""ctypes binding of libpfmscan.so (include/pfmscan.h).

The product path has no CPU fallback: if the library is missing or no gfx950
device is visible, everything here raises.  Nothing under ``oracle/`` is ever
imported from this package.
"""
import ctypes
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PFMSCAN_LIB") or os.path.join(HERE, "libpfmscan.so")   # override: A/B of two builds

OK = 0
E_BADARG, E_BADSHAPE, E_OOM, E_HIP, E_CAPACITY = -1, -2, -3, -4, -5
PROFILE_NONE, PROFILE_F32, PROFILE_F64 = 0, 1, 2
SEP = 7
NCODE = 8
NSTRUCT = 7
MAX_M = 64            # widest PFM of the tuned kernels and of PFM libraries
MAX_WIDTH = 4096      # widest PFM accepted (wider than MAX_M: the plain rolled-loop kernel)
ABI_VERSION = 8

# every symbol include/pfmscan.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "pfmscan_abi_version", "pfmscan_ctx_create", "pfmscan_ctx_destroy", "pfmscan_last_error",
    "pfmscan_device_info", "pfmscan_synchronize", "pfmscan_motif_create", "pfmscan_motif_destroy",
    "pfmscan_pwm_calculate", "pfmscan_scan_dev", "pfmscan_scan_letters_f64_dev", "pfmscan_hits_dev",
    "pfmscan_scan_host", "pfmscan_scan_letters_f64_host", "pfmscan_hits_host", "pfmscan_time_scan_dev",
    "pfmscan_stage", "pfmscan_scan_staged", "pfmscan_hits_staged", "pfmscan_hits_adaptive_dev",
    "pfmscan_library_create", "pfmscan_library_destroy", "pfmscan_library_info", "pfmscan_library_hits_dev",
    "pfmscan_library_hits_staged", "pfmscan_library_hits_host", "pfmscan_library_hits_pipeline_host", "pfmscan_debug_credit_table",
    "pfmscan_library_create_letters", "pfmscan_library_hits_letters_dev", "pfmscan_library_hits_letters_host", "pfmscan_debug_library8_credits",
    "pfmscan_debug_quad_table", "pfmscan_debug_credit8_table",
    "pfmscan_hits_pipeline_host", "pfmscan_staged_positions",
    "pfmscan_hits_letters_f64_dev", "pfmscan_hits_letters_f64_staged", "pfmscan_hits_letters_f64_host",
    "pfmscan_hits_pair_dev", "pfmscan_stage_codes2", "pfmscan_hits_pair_staged", "pfmscan_hits_pair_host", "pfmscan_round_decimals",
    "pfmscan_set_upload_mode", "pfmscan_upload_source_file", "pfmscan_upload_source_file_checked", "pfmscan_fasta_lone_cr", "pfmscan_count_bytes", "pfmscan_fasta_index", "pfmscan_fasta_ids", "pfmscan_gather_spans", "pfmscan_fasta_encode", "pfmscan_tsv_format", "pfmscan_profile_parse", "pfmscan_tsv_number",
    "pfmscan_place_alloc", "pfmscan_place_free", "pfmscan_place_note", "pfmscan_place_trim",
]
TSV_CONST, TSV_I64, TSV_F32, TSV_F64, TSV_INDEXED, TSV_FIXED, TSV_WINDOW, TSV_SPAN = range(8)


class TsvColumn(ctypes.Structure):
    """pfmscan_tsv_column of include/pfmscan.h"""
    _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32), ("data", ctypes.c_void_p),
                ("aux", ctypes.c_void_p), ("blob", ctypes.c_void_p), ("width", ctypes.c_int64)]



class CapacityError(RuntimeError):
    """Hit buffer too small; ``required`` holds the number of hits."""

    def __init__(self, msg, required):
        RuntimeError.__init__(self, msg)
        self.required = required


_lib = None


def _preload_hip_runtime():
    """A process must hold ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so / libhsa-runtime64.so; if libpfmscan (linked against /opt/rocm) came
    first, a later ``import torch`` would bring a second HSA runtime and see no GPU.  So
    when torch is installed its copy of the HIP runtime is loaded first -- by path, without
    importing torch -- and libpfmscan's NEEDED libamdhip64.so.7 binds to it."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the shared library (once).  Raises ImportError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "rnascan_amd: %s is missing -- build it with `python -m rnascan_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    _preload_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    L.pfmscan_abi_version.restype = i32
    L.pfmscan_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.pfmscan_ctx_destroy.argtypes = [vp]
    L.pfmscan_ctx_destroy.restype = None
    L.pfmscan_last_error.argtypes = [vp]
    L.pfmscan_last_error.restype = ctypes.c_char_p
    L.pfmscan_device_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i64), ctypes.c_char_p, i32]
    L.pfmscan_synchronize.argtypes = [vp]
    L.pfmscan_motif_create.argtypes = [vp, vp, vp, i32, ctypes.POINTER(vp)]
    L.pfmscan_motif_destroy.argtypes = [vp]
    L.pfmscan_motif_destroy.restype = None
    L.pfmscan_pwm_calculate.argtypes = [vp, ctypes.c_char_p, i64, vp, i64, vp]
    L.pfmscan_scan_dev.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, vp]
    L.pfmscan_scan_letters_f64_dev.argtypes = [vp, vp, vp, i64, vp, vp]
    L.pfmscan_hits_dev.argtypes = [vp, vp, vp, vp, i32, i64, dbl, dbl, i64, vp, vp, vp, vp, vp]
    L.pfmscan_hits_adaptive_dev.argtypes = [vp, vp, vp, vp, i32, i64, dbl, dbl, i64, vp, vp, vp, vp, vp]
    L.pfmscan_scan_host.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp]
    L.pfmscan_scan_letters_f64_host.argtypes = [vp, vp, vp, i64, vp]
    L.pfmscan_hits_host.argtypes = [vp, vp, vp, vp, i32, i64, dbl, dbl, i64, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_stage.argtypes = [vp, vp, vp, i32, i64]
    L.pfmscan_scan_staged.argtypes = [vp, vp, vp, vp]
    L.pfmscan_staged_positions.argtypes = [vp]
    L.pfmscan_hits_staged.argtypes = [vp, vp, dbl, dbl, i64, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_time_scan_dev.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, vp, i32, i32, ctypes.POINTER(dbl)]
    L.pfmscan_library_create.argtypes = [vp, vp, vp, i32, i32, ctypes.POINTER(vp)]
    L.pfmscan_library_create_letters.argtypes = [vp, vp, vp, i32, i32, ctypes.POINTER(vp)]
    L.pfmscan_library_hits_letters_dev.argtypes = [vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp]
    L.pfmscan_library_hits_letters_host.argtypes = [vp, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_library_destroy.argtypes = [vp]
    L.pfmscan_library_destroy.restype = None
    L.pfmscan_library_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32),
                                       ctypes.POINTER(dbl)]
    L.pfmscan_library_hits_dev.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp]
    L.pfmscan_library_hits_staged.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_library_hits_host.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, i64, vp, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_hits_pipeline_host.argtypes = [vp, vp, vp, vp, i32, i64, i64, dbl, dbl, i64, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_library_hits_pipeline_host.argtypes = [vp, vp, vp, vp, i32, i64, i64, vp, vp, i64, vp, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_debug_credit_table.argtypes = [vp, i32, dbl, i32, vp, ctypes.POINTER(dbl)]
    L.pfmscan_debug_quad_table.argtypes = [vp, i32, dbl, vp, ctypes.POINTER(dbl)]
    L.pfmscan_debug_credit8_table.argtypes = [vp, i32, dbl, vp, ctypes.POINTER(i32)]
    L.pfmscan_set_upload_mode.argtypes = [vp, i32]
    L.pfmscan_upload_source_file.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_char_p, i64]
    L.pfmscan_upload_source_file_checked.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_char_p, i64, i64, i64, i64]
    L.pfmscan_fasta_lone_cr.argtypes = [vp, i64, ctypes.POINTER(i32), i32]
    L.pfmscan_count_bytes.argtypes = [vp, i64, vp, i32]
    L.pfmscan_place_alloc.argtypes = [vp, i32, vp, vp, i32]
    L.pfmscan_place_free.argtypes = [vp, vp]
    L.pfmscan_place_trim.argtypes = [vp]
    L.pfmscan_place_note.argtypes = [vp]
    L.pfmscan_place_note.restype = ctypes.c_char_p
    L.pfmscan_fasta_index.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, ctypes.POINTER(i64), i32]
    L.pfmscan_fasta_ids.argtypes = [vp, vp, vp, i64, vp, vp, ctypes.POINTER(i32)]
    L.pfmscan_gather_spans.argtypes = [vp, vp, i64, i32, vp, i64, ctypes.POINTER(i64)]
    L.pfmscan_fasta_encode.argtypes = [vp, vp, vp, vp, i64, i64, vp, i32, vp, vp, i32]
    L.pfmscan_profile_parse.argtypes = [ctypes.c_char_p, i64, i32, i64, vp, ctypes.POINTER(i64)]
    L.pfmscan_tsv_number.argtypes = [vp, i64, i64, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i32)]
    L.pfmscan_hits_letters_f64_dev.argtypes = [vp, vp, vp, i64, dbl, i64, vp, vp, vp, vp]
    L.pfmscan_hits_letters_f64_staged.argtypes = [vp, vp, dbl, i64, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_hits_letters_f64_host.argtypes = [vp, vp, vp, i64, dbl, i64, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_hits_pair_dev.argtypes = [vp, vp, vp, vp, vp, i64, dbl, dbl, i64, vp, vp, vp, vp, vp]
    L.pfmscan_stage_codes2.argtypes = [vp, vp, i64]
    L.pfmscan_hits_pair_staged.argtypes = [vp, vp, vp, dbl, dbl, i64, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_hits_pair_host.argtypes = [vp, vp, vp, vp, vp, i64, dbl, dbl, i64, vp, vp, vp, ctypes.POINTER(i64)]
    L.pfmscan_round_decimals.argtypes = [vp, i64, i32, vp, i32]
    L.pfmscan_tsv_format.argtypes = [ctypes.POINTER(TsvColumn), i32, i64, i64, vp, i64, ctypes.POINTER(i64), vp, ctypes.POINTER(i32), i32]
    for name in SYMBOLS:          # every other entry point returns a status
        if name not in ("pfmscan_ctx_destroy", "pfmscan_motif_destroy", "pfmscan_last_error", "pfmscan_library_destroy",
                        "pfmscan_staged_positions", "pfmscan_place_note"):
            getattr(L, name).restype = i32
    L.pfmscan_staged_positions.restype = i64
    if L.pfmscan_abi_version() != ABI_VERSION:
        raise ImportError("libpfmscan ABI %d, bindings expect %d" % (L.pfmscan_abi_version(), ABI_VERSION))
    _lib = L
    return L


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    return ctypes.c_void_p(int(a))          # raw device address (e.g. torch.Tensor.data_ptr())


def is_file_mapping(a):
    """True for an array whose memory is a mapped file (numpy memmap or a view of one / of an mmap object)"""
    import mmap
    while a is not None:
        if isinstance(a, (np.memmap, mmap.mmap)):
            return True
        a = getattr(a, "base", None)
    return False


def root_memmap(a):
    """the numpy.memmap an array is (a view of) -- the one that owns the mapping and knows its file -- or None"""
    root = None
    while a is not None:
        if isinstance(a, np.memmap) and getattr(a, "filename", None) is not None:
            root = a
        a = getattr(a, "base", None)
    return root


def _raise(L, ctx, rc, n_hits=None):
    msg = L.pfmscan_last_error(ctx).decode("utf-8", "replace")
    if rc in (E_BADARG, E_BADSHAPE):
        raise ValueError(msg)               # _pwm.c:96-113 raise ValueError
    if rc == E_OOM:
        raise MemoryError(msg)              # _pwm.c:27-31
    if rc == E_CAPACITY:
        raise CapacityError(msg, n_hits)
    raise RuntimeError("libpfmscan: " + msg)


# ---------------------------------------------------------------------------
# host ingest / output (no device, no context)
# ---------------------------------------------------------------------------
def fasta_index(buf, threads=0):
    """uint8 array of FASTA bytes -> (hdr_off, hdr_len, seq_off, seq_end, n_letters), int64 [n_records] each."""
    L = load()
    buf = np.asarray(buf)
    n = ctypes.c_int64(0)
    cap = max(1024, buf.size // 64)            # untouched pages of an over-sized np.empty cost nothing
    while True:
        cols = [np.empty(cap, dtype=np.int64) for _ in range(5)]
        rc = L.pfmscan_fasta_index(_ptr(buf), buf.size, cap, *[_ptr(c) for c in cols], ctypes.byref(n), int(threads))
        if rc == E_CAPACITY and n.value > cap:
            cap = n.value                      # a file of very short records: once more with the exact count
            continue
        if rc != OK:
            _raise(L, None, rc)
        return tuple(c[:n.value] for c in cols)


def fasta_lone_cr(buf, threads=0):
    """True when the FASTA bytes hold a carriage return that is not part of \\r\\n (anywhere in the buffer)"""
    L = load()
    buf = np.asarray(buf)
    found = ctypes.c_int(0)
    rc = L.pfmscan_fasta_lone_cr(_ptr(buf), buf.size, ctypes.byref(found), int(threads))
    if rc != OK:
        _raise(L, None, rc)
    return bool(found.value)


def count_bytes(buf, threads=0):
    """int64 [256]: occurrences of every byte value in a uint8 array (numpy.bincount widens 3x10^8 bytes to int64 first: 10 s)"""
    L = load()
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    counts = np.zeros(256, dtype=np.int64)
    rc = L.pfmscan_count_bytes(_ptr(buf), buf.size, _ptr(counts), int(threads))
    if rc != OK:
        _raise(L, None, rc)
    return counts


def fasta_ids(buf, hdr_off, hdr_len):
    """(id spans int64 [n][2] = (offset, length) of every record's first header word, all headers ASCII?)"""
    L = load()
    n = int(hdr_off.size)
    off, ln = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
    ascii_ = ctypes.c_int(1)
    rc = L.pfmscan_fasta_ids(_ptr(np.asarray(buf)), _ptr(hdr_off), _ptr(hdr_len), n, _ptr(off), _ptr(ln), ctypes.byref(ascii_))
    if rc != OK:
        _raise(L, None, rc)
    return np.stack([off, ln], axis=1), bool(ascii_.value)


def gather_spans(buf, spans, separator=10):
    """bytes of the (offset, length) spans of buf, each followed by the separator byte, as one bytes object"""
    L = load()
    spans = np.ascontiguousarray(spans, dtype=np.int64)
    n = int(spans.shape[0])
    cap = int(spans[:, 1].sum()) + n if n else 0
    out = np.empty(max(cap, 1), dtype=np.uint8)
    got = ctypes.c_int64(0)
    rc = L.pfmscan_gather_spans(_ptr(np.asarray(buf)), _ptr(spans), n, int(separator), _ptr(out), cap, ctypes.byref(got))
    if rc != OK:
        _raise(L, None, rc)
    return out[:got.value].tobytes()


def fasta_encode(buf, seq_off, seq_end, n_letters, lo, hi, lut, separator=SEP, threads=0):
    """records [lo, hi) of an indexed FASTA buffer -> (codes uint8 [sum(L + 1)], offsets int64 [hi - lo])."""
    L = load()
    total = int(n_letters[lo:hi].sum()) + (hi - lo)
    codes = np.empty(total, dtype=np.uint8)
    offsets = np.empty(hi - lo, dtype=np.int64)
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    if lut.size != 256:
        raise ValueError("lut must have 256 entries")
    if hi > lo:
        rc = L.pfmscan_fasta_encode(_ptr(buf), _ptr(seq_off), _ptr(seq_end), _ptr(n_letters), lo, hi, _ptr(lut), int(separator),
                                    _ptr(codes), _ptr(offsets), int(threads))
        if rc != OK:
            _raise(L, None, rc)
    return codes, offsets


def profile_parse(data, n_cols):
    """bytes of an averaged-structure profile file -> float64 [n_rows][n_cols] (first column dropped, header skipped), the
    numbers converted exactly as pandas' read_table converts them; None when the file is one for pandas to judge"""
    L = load()
    cap = data.count(b"\n") + 1
    out = np.empty((cap, n_cols), dtype=np.float64)
    k = ctypes.c_int64(0)
    rc = L.pfmscan_profile_parse(data, len(data), int(n_cols), cap, _ptr(out), ctypes.byref(k))
    if rc == E_BADSHAPE:
        return None
    if rc != OK:
        _raise(L, None, rc)
    return out[:k.value]


def round_decimals(values, decimals=3, threads=0):
    """``[round(float(x), decimals) for x in values]`` as a float64 array: Python's float rounding (the nearest decimal
    of the EXACT binary value, ties to even -- rnascan.py:273 applies it to every reported score), natively and in
    parallel.  numpy.round is a different function (scale, rint, unscale) and differs from it near ties."""
    L = load()
    a = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty_like(a)
    rc = L.pfmscan_round_decimals(_ptr(a), a.size, int(decimals), _ptr(out), int(threads))
    if rc != OK:
        _raise(L, None, rc)
    return out


TSV_MAX_PIECES = 16


def tsv_number(block, first_id, in_quotes=0):
    """rows formatted WITHOUT their Match_ID column (bytes) -> (the rows with it, number of rows, quote state at the end of
    the block): "\\t<id>" in front of every line end outside a quoted field, ids counting up from ``first_id``"""
    L = load()
    src = np.frombuffer(block, dtype=np.uint8) if len(block) else np.zeros(1, dtype=np.uint8)
    n = len(block)
    cap = n + 21 * (block.count(b"\n") + 1) + 32
    out = np.empty(cap, dtype=np.uint8)
    n_out, n_rows, q = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(int(in_quotes))
    rc = L.pfmscan_tsv_number(_ptr(src), n, int(first_id), _ptr(out), cap, ctypes.byref(n_out), ctypes.byref(n_rows), ctypes.byref(q))
    if rc != OK:
        _raise(L, None, rc)
    return out[:n_out.value].tobytes(), int(n_rows.value), int(q.value)


def tsv_format(columns, n_rows, first_match_id=-1, threads=0, estimate=None, scratch=None):
    """columns: list of (kind, data, aux, blob, width) with numpy arrays / bytes.  Returns the rows as a list of
    memoryviews (one per formatting thread) whose concatenation is the table.
    ``scratch``: a one-element list holding a reusable uint8 array (grown here when too small) -- a writer that formats
    chunk after chunk then touches fresh pages only once; the views point into it until the next call."""
    L = load()
    keep, desc = [], (TsvColumn * len(columns))()
    for i, (kind, data, aux, blob, width) in enumerate(columns):
        ptrs = []
        for obj in (data, aux, blob):
            if obj is None:
                ptrs.append(None)
            elif isinstance(obj, (bytes, bytearray)):
                arr = np.frombuffer(obj, dtype=np.uint8) if len(obj) else np.zeros(1, dtype=np.uint8)
                keep.append(arr)
                ptrs.append(arr.ctypes.data)
            else:
                arr = np.ascontiguousarray(obj)
                keep.append(arr)
                ptrs.append(arr.ctypes.data)
        desc[i] = TsvColumn(int(kind), 0, ptrs[0], ptrs[1], ptrs[2], int(width))
    cap = int(estimate) if estimate else max(1 << 16, n_rows * 128)
    need, n_pieces = ctypes.c_int64(0), ctypes.c_int(0)
    pieces = np.zeros(2 * TSV_MAX_PIECES, dtype=np.int64)
    while True:
        if scratch is not None and scratch[0] is not None and scratch[0].size >= cap:
            out = scratch[0]
            cap = out.size
        else:
            out = np.empty(cap, dtype=np.uint8)
            if scratch is not None:
                scratch[0] = out
        rc = L.pfmscan_tsv_format(desc, len(columns), int(n_rows), int(first_match_id), _ptr(out), cap, ctypes.byref(need),
                                  _ptr(pieces), ctypes.byref(n_pieces), int(threads))
        if rc == E_CAPACITY and need.value > cap:
            cap = need.value
            continue
        if rc != OK:
            _raise(L, None, rc)
        view = memoryview(out)
        return [view[int(pieces[2 * k]):int(pieces[2 * k] + pieces[2 * k + 1])] for k in range(n_pieces.value)]


PLACE_PLAIN = 1


class PlacedArray(object):
    """one array of a set of Context.place_alloc: a raw device range that outlives this object (the set is freed by
    Context.place_free(first array) or with the context)"""

    def __init__(self, ctx, ptr, nbytes, first):
        self.ctx, self.ptr, self.nbytes, self.first = ctx, ptr, nbytes, first

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}


class Context(object):
    """One device context (one per GPU / per host thread)."""

    def __init__(self, device=0):
        self._L = load()
        h = ctypes.c_void_p()
        rc = self._L.pfmscan_ctx_create(int(device), ctypes.byref(h))
        if rc != OK:
            _raise(self._L, None, rc)
        self._h = h
        self.device = int(device)
        self._staged_n = -1
        self.scratch_gen = 0        # bumped whenever the ctx's device scratch is (re)written

    def close(self):
        if getattr(self, "_h", None):
            self._L.pfmscan_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, n_hits=None):
        if rc != OK:
            _raise(self._L, self._h, rc, n_hits)

    def device_info(self):
        n_cu, hbm = ctypes.c_int(), ctypes.c_int64()
        name = ctypes.create_string_buffer(128)
        self._check(self._L.pfmscan_device_info(self._h, ctypes.byref(n_cu), ctypes.byref(hbm), name, 128))
        return {"name": name.value.decode(), "n_cu": n_cu.value, "hbm_bytes": hbm.value}

    def synchronize(self):
        self._check(self._L.pfmscan_synchronize(self._h))

    # -- device arrays of one scan, placed together (pfmscan_place_alloc) ------------
    def place_alloc(self, sizes, plain=False):
        """device arrays of ``sizes`` bytes that one scan walks in lock-step, in parts of HBM that do not share DRAM banks
        (include/pfmscan.h: pfmscan_place_alloc); returns a list of PlacedArray (``.ptr``, ``.nbytes``, and
        ``__cuda_array_interface__``: ``torch.as_tensor(a, device=...)`` views it as uint8 without a copy)"""
        n = len(sizes)
        b = (ctypes.c_int64 * n)(*[int(x) for x in sizes])
        out = (ctypes.c_void_p * n)()
        self._check(self._L.pfmscan_place_alloc(self._h, n, b, out, PLACE_PLAIN if plain else 0))
        return [PlacedArray(self, int(out[r]), int(sizes[r]), r == 0) for r in range(n)]

    def place_free(self, first):
        self._check(self._L.pfmscan_place_free(self._h, ctypes.c_void_p(first.ptr if isinstance(first, PlacedArray) else int(first))))

    def place_note(self):
        return (self._L.pfmscan_place_note(self._h) or b"").decode()

    def place_trim(self):
        """give back the memory of the sets pfmscan_place_free keeps for reuse (include/pfmscan.h)"""
        self._check(self._L.pfmscan_place_trim(self._h))

    # -- PSSM operands ----------------------------------------------------------
    def motif(self, letter_table=None, struct_pssm=None):
        return Motif(self, letter_table, struct_pssm)

    # -- _pwm.calculate drop-in -------------------------------------------------
    def pwm_calculate(self, sequence, matrix):
        """``_pwm.calculate(sequence, matrix)`` (_pwm.c:79-121) on the GPU."""
        seq = sequence.encode("ascii") if isinstance(sequence, str) else bytes(sequence)
        M = np.asarray(matrix)
        if M.dtype != np.float64:
            raise ValueError("position-weight matrix should contain floating-point values")
        if M.ndim != 2:
            raise ValueError("position-weight matrix has incorrect rank (%d expected 2)" % M.ndim)
        if M.shape[1] != 4:
            raise ValueError("position-weight matrix should have four columns (%d columns found)" % M.shape[1])
        M = np.ascontiguousarray(M)
        n = len(seq) - M.shape[0] + 1
        if n < 0:
            raise MemoryError("failed to create output data")      # _pwm.c:26-31 on a negative shape
        out = np.empty(n, dtype=np.float32)
        self.scratch_gen += 1
        self._staged_n = -1                 # the C side restages: a following scan_staged must not trust the old length
        self._check(self._L.pfmscan_pwm_calculate(self._h, seq, len(seq), _ptr(M), M.shape[0], _ptr(out)))
        return out

    # -- host-buffer scans ------------------------------------------------------
    def scan_host(self, motif, codes, profile=None, want_seq=True, want_struct=True):
        """All window scores of a packed stream -> (seq float32[n] | None, struct float64[n] | None)."""
        n, codes, profile, dt = _stream_args(motif, codes, profile)
        out_seq = np.empty(n, dtype=np.float32) if (want_seq and motif.has_letters) else None
        out_struct = np.empty(n, dtype=np.float64) if (want_struct and motif.has_struct) else None
        self.scratch_gen += 1
        self._staged_n = -1
        self._upload_mode_for(codes, profile)
        self._check(self._L.pfmscan_scan_host(self._h, motif._h, _ptr(codes), _ptr(profile), dt, n,
                                              _ptr(out_seq), _ptr(out_struct)))
        self._staged_n = n                  # scan_host = stage + scan_staged: the stream stays staged
        return out_seq, out_struct

    def scan_letters_f64_host(self, motif, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        out = np.empty(codes.size, dtype=np.float64)
        self.scratch_gen += 1
        self._staged_n = -1
        self._check(self._L.pfmscan_scan_letters_f64_host(self._h, motif._h, _ptr(codes), codes.size, _ptr(out)))
        return out

    def hits_host(self, motif, codes, profile=None, thr_seq=-np.inf, thr_struct=-np.inf, capacity=None):
        """Thresholded hits sorted by position -> (pos int64[k], seq float32[k], struct float64[k]).
        Grows the buffer and retries when the first guess was too small."""
        n, codes, profile, dt = _stream_args(motif, codes, profile)
        cap = int(capacity) if capacity is not None else max(1024, n // 64)
        self.scratch_gen += 1
        self._staged_n = -1
        self._upload_mode_for(codes, profile)
        while True:
            pos = np.empty(cap, dtype=np.int64)
            sq = np.empty(cap, dtype=np.float32)
            st = np.empty(cap, dtype=np.float64)
            k = ctypes.c_int64(0)
            rc = self._L.pfmscan_hits_host(self._h, motif._h, _ptr(codes), _ptr(profile), dt, n,
                                           float(thr_seq), float(thr_struct), cap,
                                           _ptr(pos), _ptr(sq), _ptr(st), ctypes.byref(k))
            if rc == E_CAPACITY and capacity is None:
                cap = int(k.value)
                continue
            self._check(rc, k.value)
            k = int(k.value)
            self._staged_n = n              # hits_host = stage + hits_staged: the stream stays staged
            return pos[:k].copy(), (sq[:k].copy() if motif.has_letters else None), (st[:k].copy() if motif.has_struct else None)

    def hits_pipeline_host(self, motif, codes, profile=None, thr_seq=-np.inf, thr_struct=-np.inf, chunk_positions=0,
                           capacity=None):
        """hits_host for streams of any length (numpy arrays or memory maps): chunked, the upload of the next chunk
        overlaps the scan of the current one, device scratch = two chunks.  Same return values as hits_host."""
        n, codes, profile, dt = _stream_args(motif, codes, profile)
        cap = int(capacity) if capacity is not None else max(1024, n // 64)
        self.scratch_gen += 1
        self._staged_n = -1
        self._upload_mode_for(codes, profile)
        while True:
            pos = np.empty(cap, dtype=np.int64)
            sq = np.empty(cap, dtype=np.float32)
            st = np.empty(cap, dtype=np.float64)
            k = ctypes.c_int64(0)
            rc = self._L.pfmscan_hits_pipeline_host(self._h, motif._h, _ptr(codes), _ptr(profile), dt, n, int(chunk_positions),
                                                    float(thr_seq), float(thr_struct), cap, _ptr(pos), _ptr(sq), _ptr(st),
                                                    ctypes.byref(k))
            if rc == E_CAPACITY and capacity is None:
                cap = int(k.value)
                continue
            self._check(rc, k.value)
            k = int(k.value)
            return pos[:k].copy(), (sq[:k].copy() if motif.has_letters else None), (st[:k].copy() if motif.has_struct else None)

    def _upload_mode_for(self, *arrays):
        """staged uploads (parallel copy into pinned buffers) for sources that are file mappings -- a packed profile
        store -- and the runtime's in-place pinning for ordinary memory (include/pfmscan.h, pfmscan_set_upload_mode)"""
        mode = 1 if any(is_file_mapping(a) for a in arrays) else 0
        if mode != getattr(self, "_upload_mode", 0):
            self._check(self._L.pfmscan_set_upload_mode(self._h, mode))
            self._upload_mode = mode
        if mode:
            for a in arrays:
                self._register_mapping(root_memmap(a))

    def _register_mapping(self, mm):
        """tell the uploader which file a read-only numpy.memmap maps (pfmscan_upload_source_file): staged transfers then
        pread the file and never touch the mapping (no page faults, nothing to unmap page by page at exit).  The range is
        forgotten again just before the memmap goes away."""
        import weakref
        if mm is None or getattr(mm, "mode", None) != "r" or mm.nbytes == 0:      # "c": private changes are not in the file
            return
        known = self.__dict__.setdefault("_mappings", {})
        base = mm.ctypes.data
        if base in known:
            return
        # which file was mapped: recorded by whoever opened the memmap (store.ProfileStore), else taken now -- the checked
        # form refuses a path that has come to name another file since (a re-packed store), uploads then read the mapping
        ident = getattr(mm, "_mapped_file_id", None)
        if ident is None:
            try:
                st = os.stat(mm.filename)
                ident = (st.st_dev, st.st_ino, st.st_size)
            except OSError:
                return
        rc = self._L.pfmscan_upload_source_file_checked(self._h, ctypes.c_void_p(base), mm.nbytes, os.fsencode(str(mm.filename)),
                                                        int(mm.offset), int(ident[0]), int(ident[1]), int(ident[2]))
        if rc:
            return                                      # not fatal: the mapping itself is still a valid source
        ctx_ref = weakref.ref(self)

        def forget(base=base):
            ctx = ctx_ref()
            if ctx is not None and getattr(ctx, "_h", None):
                ctx._L.pfmscan_upload_source_file(ctx._h, ctypes.c_void_p(base), 0, None, 0)
                ctx.__dict__.get("_mappings", {}).pop(base, None)
        known[base] = weakref.finalize(mm, forget)

    # -- staged stream: upload once, run many motifs ------------------------------------
    def stage(self, codes=None, profile=None):
        """copy a packed stream into the ctx's device scratch; returns n_pos"""
        n = None
        dt = PROFILE_NONE
        if codes is not None:
            codes = np.ascontiguousarray(codes, dtype=np.uint8)
            n = codes.size
        if profile is not None:
            if profile.dtype == np.float32:
                dt = PROFILE_F32
            elif profile.dtype == np.float64:
                dt = PROFILE_F64
            else:
                raise ValueError("profile must be float32 or float64")
            profile = np.ascontiguousarray(profile)
            if profile.ndim != 2 or profile.shape[1] != NSTRUCT:
                raise ValueError("profile must be [n_pos][7]")
            if n is not None and profile.shape[0] != n:
                raise ValueError("codes and profile disagree on n_pos")
            n = profile.shape[0]
        if n is None:
            raise ValueError("nothing to stage")
        self.scratch_gen += 1
        self._upload_mode_for(codes, profile)
        self._check(self._L.pfmscan_stage(self._h, _ptr(codes), _ptr(profile), dt, n))
        self._staged_n = n
        return self.scratch_gen

    def scan_staged(self, motif, want_seq=True, want_struct=True):
        n = int(self._L.pfmscan_staged_positions(self._h))      # the library's own count sizes the outputs, not a Python copy of it
        if n < 0:
            raise ValueError("no stream staged (call stage first)")
        out_seq = np.empty(n, dtype=np.float32) if (want_seq and motif.has_letters) else None
        out_struct = np.empty(n, dtype=np.float64) if (want_struct and motif.has_struct) else None
        self._check(self._L.pfmscan_scan_staged(self._h, motif._h, _ptr(out_seq), _ptr(out_struct)))
        return out_seq, out_struct

    def hits_staged(self, motif, thr_seq=-np.inf, thr_struct=-np.inf, capacity=None):
        n = int(self._L.pfmscan_staged_positions(self._h))
        if n < 0:
            raise ValueError("no stream staged (call stage first)")
        cap = int(capacity) if capacity is not None else max(1024, n // 64)
        while True:
            pos = np.empty(cap, dtype=np.int64)
            sq = np.empty(cap, dtype=np.float32)
            st = np.empty(cap, dtype=np.float64)
            k = ctypes.c_int64(0)
            rc = self._L.pfmscan_hits_staged(self._h, motif._h, float(thr_seq), float(thr_struct), cap,
                                             _ptr(pos), _ptr(sq), _ptr(st), ctypes.byref(k))
            if rc == E_CAPACITY and capacity is None:
                cap = int(k.value)
                continue
            self._check(rc, k.value)
            k = int(k.value)
            return pos[:k].copy(), (sq[:k].copy() if motif.has_letters else None), (st[:k].copy() if motif.has_struct else None)

    # -- generic-alphabet letter hits in fp64; two code streams ---------------------------------
    def hits_letters_f64_staged(self, motif, thr, capacity=None):
        """hits of a letters-only motif over the staged codes, fp64 compare and score (matrix.py:25-43 + rnascan.py:263)
        -> (pos int64[k] sorted, score float64[k])"""
        n = int(self._L.pfmscan_staged_positions(self._h))
        if n < 0:
            raise ValueError("no stream staged (call stage first)")
        cap = int(capacity) if capacity is not None else max(1024, n // 64)
        while True:
            pos = np.empty(cap, dtype=np.int64)
            sc = np.empty(cap, dtype=np.float64)
            k = ctypes.c_int64(0)
            rc = self._L.pfmscan_hits_letters_f64_staged(self._h, motif._h, float(thr), cap, _ptr(pos), _ptr(sc), ctypes.byref(k))
            if rc == E_CAPACITY and capacity is None:
                cap = int(k.value)
                continue
            self._check(rc, k.value)
            k = int(k.value)
            return pos[:k].copy(), sc[:k].copy()

    def hits_letters_f64_host(self, motif, codes, thr, capacity=None):
        self.stage(codes, None)
        return self.hits_letters_f64_staged(motif, thr, capacity)

    def stage_codes2(self, codes2):
        """a second code stream (the structure letters of the staged records) beside the staged one"""
        codes2 = np.ascontiguousarray(codes2, dtype=np.uint8)
        self._check(self._L.pfmscan_stage_codes2(self._h, _ptr(codes2), codes2.size))

    def hits_pair_staged(self, motif_seq, motif_struct, thr_seq, thr_struct, capacity=None):
        """hits of (sequence letters AND structure letters) over the two staged code streams
        -> (pos int64[k] sorted, seq float32[k], struct float64[k])"""
        n = int(self._L.pfmscan_staged_positions(self._h))
        if n < 0:
            raise ValueError("no stream staged (call stage first)")
        cap = int(capacity) if capacity is not None else max(1024, n // 64)
        while True:
            pos = np.empty(cap, dtype=np.int64)
            sq = np.empty(cap, dtype=np.float32)
            st = np.empty(cap, dtype=np.float64)
            k = ctypes.c_int64(0)
            rc = self._L.pfmscan_hits_pair_staged(self._h, motif_seq._h, motif_struct._h, float(thr_seq), float(thr_struct), cap,
                                                  _ptr(pos), _ptr(sq), _ptr(st), ctypes.byref(k))
            if rc == E_CAPACITY and capacity is None:
                cap = int(k.value)
                continue
            self._check(rc, k.value)
            k = int(k.value)
            return pos[:k].copy(), sq[:k].copy(), st[:k].copy()

    def hits_pair_host(self, motif_seq, motif_struct, codes, codes2, thr_seq, thr_struct, capacity=None):
        self.stage(codes, None)
        self.stage_codes2(codes2)
        return self.hits_pair_staged(motif_seq, motif_struct, thr_seq, thr_struct, capacity)

    def hits_letters_f64_dev(self, motif, d_codes, n_pos, thr, capacity, d_hit_pos, d_hit_score, d_hit_count, stream=None):
        self._check(self._L.pfmscan_hits_letters_f64_dev(self._h, motif._h, _ptr(d_codes), int(n_pos), float(thr), int(capacity),
                                                         _ptr(d_hit_pos), _ptr(d_hit_score), _ptr(d_hit_count), _ptr(stream)))

    def hits_pair_dev(self, motif_seq, motif_struct, d_codes, d_codes2, n_pos, thr_seq, thr_struct, capacity,
                      d_hit_pos, d_hit_seq, d_hit_struct, d_hit_count, stream=None):
        self._check(self._L.pfmscan_hits_pair_dev(self._h, motif_seq._h, motif_struct._h, _ptr(d_codes), _ptr(d_codes2), int(n_pos),
                                                  float(thr_seq), float(thr_struct), int(capacity), _ptr(d_hit_pos), _ptr(d_hit_seq),
                                                  _ptr(d_hit_struct), _ptr(d_hit_count), _ptr(stream)))

    # -- multi-PFM libraries (every motif in one pass) --------------------------------
    def library(self, letter_tables, struct_pssms=None, struct_letters=None):
        return Library(self, letter_tables, struct_pssms, struct_letters)

    def library_hits_letters_dev(self, lib, d_codes, d_codes2, n_pos, thr_seq, thr_struct, capacity,
                                 d_hit_pos, d_hit_motif, d_hit_seq, d_hit_struct, d_hit_count, stream=None):
        """LETTER library on device-resident code streams (d_codes2 None for a structure-letter library): asynchronous on
        ``stream``; hits unordered, *d_hit_count = total (above capacity = incomplete)"""
        ts, tt = lib.thresholds(thr_seq, thr_struct)
        self._check(self._L.pfmscan_library_hits_letters_dev(self._h, lib._h, _ptr(d_codes), _ptr(d_codes2), int(n_pos), _ptr(ts), _ptr(tt),
                                                             int(capacity), _ptr(d_hit_pos), _ptr(d_hit_motif), _ptr(d_hit_seq),
                                                             _ptr(d_hit_struct), _ptr(d_hit_count), _ptr(stream)))

    def library_hits_letters_host(self, lib, codes, codes2=None, thr_seq=None, thr_struct=None, capacity=None):
        """hits of a LETTER library (Library(struct_letters=...)): ``codes`` is the 8-code stream of a structure-letter library,
        or the sequence codes of a two-FASTA library whose structure strings are ``codes2`` -> as library_hits_staged"""
        if not lib.letter_kind:
            raise ValueError("not a letter library")
        self.stage(codes, None)
        if lib.has_letters:
            self.stage_codes2(codes2)
        return self.library_hits_staged(lib, thr_seq, thr_struct, capacity)

    def library_hits_staged(self, lib, thr_seq, thr_struct=None, capacity=None):
        """hits of every motif of ``lib`` over the staged stream, sorted by (position, motif index)
        -> (pos int64[k], motif int32[k], seq float32[k], struct float64[k] | None)"""
        n = int(self._L.pfmscan_staged_positions(self._h))
        if n < 0:
            raise ValueError("no stream staged (call stage first)")
        ts, tt = lib.thresholds(thr_seq, thr_struct)
        cap = int(capacity) if capacity is not None else max(4096, n // 16)
        while True:
            pos = np.empty(cap, dtype=np.int64)
            mo = np.empty(cap, dtype=np.int32)
            sq = np.empty(cap, dtype=np.float32)
            st = np.empty(cap, dtype=np.float64)
            k = ctypes.c_int64(0)
            rc = self._L.pfmscan_library_hits_staged(self._h, lib._h, _ptr(ts), _ptr(tt), cap, _ptr(pos), _ptr(mo),
                                                     _ptr(sq), _ptr(st), ctypes.byref(k))
            if rc == E_CAPACITY and capacity is None:
                cap = int(k.value)
                continue
            self._check(rc, k.value)
            k = int(k.value)
            return pos[:k].copy(), mo[:k].copy(), (sq[:k].copy() if lib.has_letters else None), (st[:k].copy() if lib.has_struct else None)

    def library_hits_host(self, lib, codes, profile=None, thr_seq=None, thr_struct=None, capacity=None):
        self.stage(codes if lib.has_letters else None, profile if lib.has_struct else None)
        return self.library_hits_staged(lib, thr_seq, thr_struct, capacity)

    def library_hits_pipeline_host(self, lib, codes, profile=None, thr_seq=None, thr_struct=None, chunk_positions=0, capacity=None):
        """library_hits_host for host streams of any length (numpy arrays or memory maps): chunked, the upload of the next
        chunk overlaps the scan of the current one, device scratch = two chunks; nothing stays staged"""
        codes = None if (codes is None or not lib.has_letters) else np.ascontiguousarray(codes, dtype=np.uint8)
        dt = PROFILE_NONE
        if lib.has_struct:
            if profile is None:
                raise ValueError("library has structure PSSMs: profile required")
            if profile.dtype == np.float32:
                dt = PROFILE_F32
            elif profile.dtype == np.float64:
                dt = PROFILE_F64
            else:
                raise ValueError("profile must be float32 or float64")
            profile = np.ascontiguousarray(profile)
            if profile.ndim != 2 or profile.shape[1] != NSTRUCT:
                raise ValueError("profile must be [n_pos][7]")
        else:
            profile = None
        n = int(codes.size if codes is not None else profile.shape[0])
        if codes is not None and profile is not None and profile.shape[0] != n:
            raise ValueError("codes and profile disagree on n_pos")
        ts, tt = lib.thresholds(thr_seq, thr_struct)
        cap = int(capacity) if capacity is not None else max(4096, n // 16)
        self.scratch_gen += 1
        self._staged_n = -1
        self._upload_mode_for(codes, profile)
        while True:
            pos = np.empty(cap, dtype=np.int64)
            mo = np.empty(cap, dtype=np.int32)
            sq = np.empty(cap, dtype=np.float32)
            st = np.empty(cap, dtype=np.float64)
            k = ctypes.c_int64(0)
            rc = self._L.pfmscan_library_hits_pipeline_host(self._h, lib._h, _ptr(codes), _ptr(profile), dt, n, int(chunk_positions),
                                                            _ptr(ts), _ptr(tt), cap, _ptr(pos), _ptr(mo), _ptr(sq), _ptr(st),
                                                            ctypes.byref(k))
            if rc == E_CAPACITY and capacity is None:
                cap = int(k.value)
                continue
            self._check(rc, k.value)
            k = int(k.value)
            return pos[:k].copy(), mo[:k].copy(), (sq[:k].copy() if lib.has_letters else None), (st[:k].copy() if lib.has_struct else None)

    def library_hits_dev(self, lib, d_codes, d_profile, profile_dtype, n_pos, thr_seq, thr_struct, capacity,
                         d_hit_pos, d_hit_motif, d_hit_seq, d_hit_struct, d_hit_count, stream=None):
        """asynchronous on ``stream``; hits unordered, *d_hit_count = total (above capacity = incomplete)"""
        ts, tt = lib.thresholds(thr_seq, thr_struct)
        self._check(self._L.pfmscan_library_hits_dev(self._h, lib._h, _ptr(d_codes), _ptr(d_profile), int(profile_dtype),
                                                     int(n_pos), _ptr(ts), _ptr(tt), int(capacity), _ptr(d_hit_pos),
                                                     _ptr(d_hit_motif), _ptr(d_hit_seq), _ptr(d_hit_struct),
                                                     _ptr(d_hit_count), _ptr(stream)))

    # -- device-pointer scans (pointers are ints, e.g. torch data_ptr()) ----------
    def scan_dev(self, motif, d_codes, d_profile, profile_dtype, n_pos, d_out_seq, d_out_struct, stream=None):
        self._check(self._L.pfmscan_scan_dev(self._h, motif._h, _ptr(d_codes), _ptr(d_profile), int(profile_dtype),
                                             int(n_pos), _ptr(d_out_seq), _ptr(d_out_struct), _ptr(stream)))

    def scan_letters_f64_dev(self, motif, d_codes, n_pos, d_out, stream=None):
        self._check(self._L.pfmscan_scan_letters_f64_dev(self._h, motif._h, _ptr(d_codes), int(n_pos), _ptr(d_out), _ptr(stream)))

    def hits_dev(self, motif, d_codes, d_profile, profile_dtype, n_pos, thr_seq, thr_struct, capacity,
                 d_hit_pos, d_hit_seq, d_hit_struct, d_hit_count, stream=None):
        self._check(self._L.pfmscan_hits_dev(self._h, motif._h, _ptr(d_codes), _ptr(d_profile), int(profile_dtype),
                                             int(n_pos), float(thr_seq), float(thr_struct), int(capacity),
                                             _ptr(d_hit_pos), _ptr(d_hit_seq), _ptr(d_hit_struct), _ptr(d_hit_count),
                                             _ptr(stream)))

    def hits_adaptive_dev(self, motif, d_codes, d_profile, profile_dtype, n_pos, thr_seq, thr_struct, capacity,
                          d_hit_pos, d_hit_seq, d_hit_struct, d_hit_count, stream=None):
        """hits_dev that may run candidate-then-verify (synchronises the stream)"""
        self._check(self._L.pfmscan_hits_adaptive_dev(self._h, motif._h, _ptr(d_codes), _ptr(d_profile),
                                                      int(profile_dtype), int(n_pos), float(thr_seq), float(thr_struct),
                                                      int(capacity), _ptr(d_hit_pos), _ptr(d_hit_seq), _ptr(d_hit_struct),
                                                      _ptr(d_hit_count), _ptr(stream)))

    def time_scan_dev(self, motif, d_codes, d_profile, profile_dtype, n_pos, d_out_seq, d_out_struct,
                      stream=None, warmup=1, iters=5):
        ms = ctypes.c_double(0.0)
        self._check(self._L.pfmscan_time_scan_dev(self._h, motif._h, _ptr(d_codes), _ptr(d_profile), int(profile_dtype),
                                                  int(n_pos), _ptr(d_out_seq), _ptr(d_out_struct), _ptr(stream),
                                                  int(warmup), int(iters), ctypes.byref(ms)))
        return ms.value


class Motif(object):
    """Device-resident PSSM operands: letter table [m][8] and/or structure PSSM [m][7]."""

    def __init__(self, ctx, letter_table=None, struct_pssm=None):
        self._ctx = ctx
        self._L = ctx._L
        lt = None if letter_table is None else np.ascontiguousarray(letter_table, dtype=np.float64)
        sp = None if struct_pssm is None else np.ascontiguousarray(struct_pssm, dtype=np.float64)
        if lt is not None and (lt.ndim != 2 or lt.shape[1] != NCODE):
            raise ValueError("letter_table must be [m][8]")
        if sp is not None and (sp.ndim != 2 or sp.shape[1] != NSTRUCT):
            raise ValueError("struct_pssm must be [m][7]")
        if lt is None and sp is None:
            raise ValueError("motif needs a letter table and/or a structure PSSM")
        if lt is not None and sp is not None and lt.shape[0] != sp.shape[0]:
            raise ValueError("sequence and structure PFMs must have the same width for a combined scan")
        self.m = int((lt if lt is not None else sp).shape[0])
        self.has_letters = lt is not None
        self.has_struct = sp is not None
        h = ctypes.c_void_p()
        ctx._check(self._L.pfmscan_motif_create(ctx._h, _ptr(lt), _ptr(sp), self.m, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._L.pfmscan_motif_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Library(object):
    """n motifs of one width resident on the device for one-pass scans:
    letter_tables [n][m][8] (4-letter alphabet), struct_pssms [n][m][7] or None.
    letter_tables None + struct_pssms: a STRUCTURE-ONLY library (k_profile_lib: every motif in one pass over the profile).
    ``struct_letters`` [n][m][8] (up to 7 letters, fp64 scores) instead of struct_pssms: a LETTER library --
    alone a structure-letter library over one 8-code stream (k_library8), with letter_tables a two-FASTA library over two
    code streams (the structure letters of k_library's survivors)."""

    def __init__(self, ctx, letter_tables, struct_pssms=None, struct_letters=None):
        self._ctx = ctx
        self._L = ctx._L
        lt = None if letter_tables is None else np.ascontiguousarray(letter_tables, dtype=np.float64)
        if lt is not None and (lt.ndim != 3 or lt.shape[2] != NCODE):
            raise ValueError("letter_tables must be [n][m][8]")
        sp = None
        self.letter_kind = struct_letters is not None
        if self.letter_kind:
            if struct_pssms is not None:
                raise ValueError("struct_pssms and struct_letters exclude each other")
            sp = np.ascontiguousarray(struct_letters, dtype=np.float64)
            if sp.ndim != 3 or sp.shape[2] != NCODE or (lt is not None and sp.shape[:2] != lt.shape[:2]):
                raise ValueError("struct_letters must be [n][m][8] with the letter tables' n and m")
        elif struct_pssms is not None:
            sp = np.ascontiguousarray(struct_pssms, dtype=np.float64)
            if sp.ndim != 3 or sp.shape[2] != NSTRUCT or (lt is not None and sp.shape[:2] != lt.shape[:2]):
                raise ValueError("struct_pssms must be [n][m][7] with the letter tables' n and m")
        if lt is None and sp is None:
            raise ValueError("library needs letter tables and/or structure PSSMs")
        shape = (lt if lt is not None else sp).shape
        self.n, self.m = int(shape[0]), int(shape[1])
        self.has_letters = lt is not None
        self.has_struct = sp is not None
        h = ctypes.c_void_p()
        if self.letter_kind:
            ctx._check(self._L.pfmscan_library_create_letters(ctx._h, _ptr(lt), _ptr(sp), self.n, self.m, ctypes.byref(h)))
        else:
            ctx._check(self._L.pfmscan_library_create(ctx._h, _ptr(lt), _ptr(sp), self.n, self.m, ctypes.byref(h)))
        self._h = h

    def thresholds(self, thr_seq, thr_struct=None):
        """scalars or per-motif arrays -> float64 [n] arrays (None for a side the library does not have)"""
        ts = None
        if self.has_letters:
            ts = np.ascontiguousarray(np.broadcast_to(np.asarray(thr_seq, dtype=np.float64), (self.n,)))
        tt = None
        if self.has_struct:
            tt = np.ascontiguousarray(np.broadcast_to(np.asarray(-np.inf if thr_struct is None else thr_struct,
                                                                 dtype=np.float64), (self.n,)))
        return ts, tt

    def info(self):
        n, m, npass, per, eps = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        self._ctx._check(self._L.pfmscan_library_info(self._h, ctypes.byref(n), ctypes.byref(m), ctypes.byref(npass),
                                                      ctypes.byref(per), ctypes.byref(eps)))
        return {"n_motifs": n.value, "m": m.value, "passes": npass.value, "motifs_per_pass": per.value,
                "max_prefilter_eps": eps.value}

    def close(self):
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._L.pfmscan_library_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def credit_table(letter_table, thr_seq, bits=16):
    """Host-only diagnostic (no device needed): the unsigned two-letter credit table the integer prefilters use for
    ONE motif at threshold ``thr_seq`` -> (credits uint16 [ceil(m/2)][16], slack in score units).  ``bits`` = 16, 10
    (the library kernel's twelve-motifs-per-entry form for widths up to 16) or 0 (what the library kernel uses at
    this width).  A window whose credits sum (mod 2**bits) has bit bits - 1 clear cannot be a hit; tests check that
    exhaustively."""
    L = load()
    T = np.ascontiguousarray(letter_table, dtype=np.float64)
    if T.ndim != 2 or T.shape[1] != NCODE:
        raise ValueError("letter_table must be [m][8]")
    m = T.shape[0]
    out = np.zeros(((m + 1) // 2, 16), dtype=np.uint16)
    slack = ctypes.c_double(0.0)
    rc = L.pfmscan_debug_credit_table(_ptr(T), m, float(thr_seq), int(bits), _ptr(out), ctypes.byref(slack))
    if rc != OK:
        raise ValueError("pfmscan_debug_credit_table: bad argument")
    return out, slack.value


def quad_table(letter_table, thr_seq):
    """Host-only diagnostic: the four-letter credit table of k_letters_quad for ONE motif (width <= 32) at ``thr_seq``
    -> (credits uint16 [ceil(m/4)][256], slack in score units).  A window whose credits sum (mod 2**16) has bit 15 clear
    cannot be a hit."""
    L = load()
    T = np.ascontiguousarray(letter_table, dtype=np.float64)
    if T.ndim != 2 or T.shape[1] != NCODE:
        raise ValueError("letter_table must be [m][8]")
    m = T.shape[0]
    out = np.zeros(((m + 3) // 4, 256), dtype=np.uint16)
    slack = ctypes.c_double(0.0)
    rc = L.pfmscan_debug_quad_table(_ptr(T), m, float(thr_seq), _ptr(out), ctypes.byref(slack))
    if rc != OK:
        raise ValueError("pfmscan_debug_quad_table: bad argument")
    return out, slack.value


def library8_credits(letter_table, thr):
    """Host-only diagnostic: the single-letter credits k_library8 uses for ONE motif of a structure-letter library
    -> (credits uint16 [4 ceil(m/4)][8], slack in score units; inf: no prefilter for this motif)"""
    L = load()
    T = np.ascontiguousarray(letter_table, dtype=np.float64)
    if T.ndim != 2 or T.shape[1] != NCODE:
        raise ValueError("letter_table must be [m][8]")
    rows = (T.shape[0] + 3) // 4 * 4
    out = np.zeros((rows, 8), dtype=np.uint16)
    slack = ctypes.c_double(0.0)
    L.pfmscan_debug_library8_credits.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    rc = L.pfmscan_debug_library8_credits(_ptr(T), T.shape[0], float(thr), _ptr(out), ctypes.byref(slack))
    if rc != OK:
        raise ValueError("pfmscan_debug_library8_credits: bad argument")
    return out, slack.value


def credit8_table(letter_table, thr):
    """Host-only diagnostic: the single-letter credit table of k_letters_cred8 for ONE generic-alphabet motif (width <= 32)
    -> (credits uint16 [m][8], mode).  mode 1: used; 2: dense threshold (exact kernel instead); 3: no prefilter possible."""
    L = load()
    T = np.ascontiguousarray(letter_table, dtype=np.float64)
    if T.ndim != 2 or T.shape[1] != NCODE:
        raise ValueError("letter_table must be [m][8]")
    out = np.zeros((T.shape[0], 8), dtype=np.uint16)
    mode = ctypes.c_int(0)
    rc = L.pfmscan_debug_credit8_table(_ptr(T), T.shape[0], float(thr), _ptr(out), ctypes.byref(mode))
    if rc != OK:
        raise ValueError("pfmscan_debug_credit8_table: bad argument")
    return out, mode.value


def _stream_args(motif, codes, profile):
    if motif.has_letters:
        if codes is None:
            raise ValueError("motif has a letter table: codes required")
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        n = codes.size
    dt = PROFILE_NONE
    if motif.has_struct:
        if profile is None:
            raise ValueError("motif has a structure PSSM: profile required")
        if profile.dtype == np.float32:
            dt = PROFILE_F32
        elif profile.dtype == np.float64:
            dt = PROFILE_F64
        else:
            raise ValueError("profile must be float32 or float64")
        profile = np.ascontiguousarray(profile)
        if profile.ndim != 2 or profile.shape[1] != NSTRUCT:
            raise ValueError("profile must be [n_pos][7]")
        if motif.has_letters and profile.shape[0] != n:
            raise ValueError("codes and profile disagree on n_pos")
        n = profile.shape[0]
    else:
        profile = None
    return n, (codes if motif.has_letters else None), profile, dt
