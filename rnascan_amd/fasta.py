"""FASTA / background / averaged-structure file handling (host side, no Biopython).

Mirrors, by behaviour, the data-prep functions of rnascan/rnascan.py:
``parse_sequences`` (:170-174), ``preprocess_seq`` (:177-204),
``compute_background`` (:440-465), ``load_background`` (:468-484) and the
profile reading at the top of ``scan_averaged_structure`` (:296-297).
"""
import ast
import bz2
import glob
import gzip
import mmap
import os
import re
import sys
import warnings
from collections import namedtuple

import numpy as np

from . import pack

Record = namedtuple("Record", ["id", "description", "seq"])

RNA = "GAUC"                       # IUPACUnambiguousRNA.letters (Biopython order)
STRUCT = pack.STRUCT_LETTERS       # ContextualSecondaryStructure.letters


def is_rna_letters(letters):
    """True for the RNA alphabet (the reference tests isinstance(alphabet, IUPACAmbiguousRNA))."""
    return set(letters) == set(RNA)


def eprint(*args, **kwargs):
    print(*args, file=sys.stderr, **kwargs)


def _open(path):
    """fileinput.hook_compressed: by extension .gz / .bz2, else plain text."""
    ext = os.path.splitext(path)[1]
    if ext == ".gz":
        return gzip.open(path, "rt")
    if ext == ".bz2":
        return bz2.open(path, "rt")
    return open(path, "r")


def parse_sequences(fasta_files):
    """Iterate FASTA records over one file or a list of files (SeqIO 'fasta'
    semantics: id = first word of the header, description = whole header)."""
    if isinstance(fasta_files, str):
        fasta_files = [fasta_files]
    for path in fasta_files:
        with _open(path) as fh:
            header, chunks = None, []
            for line in fh:
                if line.startswith(">"):
                    if header is not None:
                        yield _record(header, chunks)
                    header, chunks = line[1:].rstrip("\r\n"), []
                elif header is not None:
                    chunks.append(line.strip())
            if header is not None:
                yield _record(header, chunks)


class _Headers(object):
    """the headers (or ids) of an indexed FASTA, decoded when asked for: a sequence-only scan needs the strings
    of the records that have hits, not 10^5 of them up front"""

    def __init__(self, owner, want_id):
        self.owner = owner
        self.want_id = want_id
        self.cache = owner._header_cache

    def __len__(self):
        return len(self.owner)

    def _one(self, i):
        got = self.cache.get(i)
        if got is None:
            header = self.owner._header(i)
            words = header.split(None, 1)
            got = self.cache[i] = (words[0] if words else "", header)
        return got[0] if self.want_id else got[1]

    def tolist(self, lo=0, hi=None):
        """items [lo, hi) as a list; ASCII headers of a plain file come over in one native gather + one split"""
        hi = len(self) if hi is None else hi
        bulk = self.owner._bulk_strings(self.want_id)
        if bulk is not None:
            return bulk[lo:hi]
        return [self._one(i) for i in range(lo, hi)]

    def __getitem__(self, key):
        if isinstance(key, slice):
            lo, hi, step = key.indices(len(self))
            return self.tolist(lo, hi)[::step] if step != 1 else self.tolist(lo, hi)
        if key < 0:
            key += len(self)
        return self._one(key)

    def __iter__(self):
        return iter(self.tolist())


class _View(object):
    """items [lo, hi) of a lazily decoded sequence, still lazy"""

    def __init__(self, seq, lo, hi):
        self.seq, self.lo, self.hi = seq, lo, hi

    def __len__(self):
        return self.hi - self.lo

    def __getitem__(self, i):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self.seq[self.lo + i]

    def __iter__(self):
        if hasattr(self.seq, "tolist"):
            return iter(self.seq.tolist(self.lo, self.hi))
        return iter([self.seq[i] for i in range(self.lo, self.hi)])


class FastaSlice(object):
    """records [lo, hi) of a LazyFasta: a sequence of Records (parsed when iterated) that can also hand over its
    letters already in stream form (``pack_rna``) without making a Python object per record."""

    def __init__(self, parent, lo, hi):
        self.parent, self.lo, self.hi = parent, lo, hi

    def __len__(self):
        return self.hi - self.lo

    def __iter__(self):
        return iter(self.parent._read(self.lo, self.hi))

    def __getitem__(self, key):
        if isinstance(key, slice):
            lo, hi, step = key.indices(len(self))
            if step != 1:
                raise ValueError("LazyFasta slices are contiguous")
            return FastaSlice(self.parent, self.lo + lo, self.lo + max(lo, hi))
        if key < 0:
            key += len(self)
        return self.parent._read(self.lo + key, self.lo + key + 1)[0]

    @property
    def ids(self):
        return _View(self.parent.ids, self.lo, self.hi)

    @property
    def descriptions(self):
        return _View(self.parent.headers, self.lo, self.hi)

    def span_tables(self):
        """(file bytes, id spans [n][2], header spans [n][2]) when the slice sits in ONE plain file whose headers are
        ASCII -- the writer then copies ids and descriptions from the mapped file itself -- else None."""
        return self.parent._span_tables(self.lo, self.hi)

    def pack_rna(self):
        """(codes uint8 with one separator after every record, offsets, lengths) of the slice, the letters mapped
        as preprocess_seq + pack.encode_rna map them (rnascan.py:186-197, _pwm.c:41-63: case-insensitive, T = U,
        everything else foreign) -- by pfmscan_fasta_encode; None when a record does not sit in a plain file."""
        return self.parent._pack(self.lo, self.hi, pack._RNA_LUT)

    def pack_letters(self, lut):
        """the same for any 256-entry letter LUT (a generic alphabet: pack.letter_lut): no transcription, no upper-casing
        (rnascan.py:186-197 leaves structure strings as they are)"""
        return self.parent._pack(self.lo, self.hi, lut)


class LazyFasta(object):
    """A FASTA file as a sliceable sequence of Records that holds only an INDEX in memory: header location, byte range
    and letter count per record, from one native pass over the mapped bytes (``pfmscan_fasta_index``).  ``lazy[a:b]``
    is a FastaSlice: a rank of a sharded run (shard.scan_sharded) reads only its own share and a batch only its own
    records -- the reference hands every record to a pool worker through one iterator (rnascan.py:379-395).
    Compressed input (.gz / .bz2) cannot be mapped: its records are parsed once and kept.  So are files with CR-only line
    ends (old Mac): the native index breaks lines at \\n only, the reference's text-mode reader at any of \\n, \\r\\n, \\r."""

    def __init__(self, fasta_files):
        from . import _lib
        self.files = [fasta_files] if isinstance(fasta_files, str) else list(fasta_files)
        self._header_cache = {}
        self._spans = {}                       # per file: (id spans, header spans) or None (non-ASCII headers)
        self._bulk = {}                        # "ids" / "headers" -> list of every record's string, or None
        self._maps = []                        # per file: the mmap object (bytes slices for the headers)
        self._bufs = []                        # per file: uint8 view of the mapped bytes, or None (compressed)
        self._index = []                       # per file: (hdr_off, hdr_len, seq_off, seq_end, n_letters)
        self._parsed = {}                      # global record index -> Record (compressed files)
        file_of, local_of, lengths = [], [], []
        for fi, path in enumerate(self.files):
            mm = buf = None
            if os.path.splitext(path)[1] not in (".gz", ".bz2"):
                if os.path.getsize(path):
                    with open(path, "rb") as fh:
                        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
                    buf = np.frombuffer(mm, dtype=np.uint8)
                    # a carriage return that is not part of \r\n ANYWHERE in the file (one native pass at memory speed):
                    # universal newlines make it a line end, the native index would not -- such a file is parsed instead
                    if _lib.fasta_lone_cr(buf):
                        mm = buf = None
                else:
                    mm, buf = b"", np.zeros(0, dtype=np.uint8)
            if buf is None:
                self._maps.append(None)
                self._bufs.append(None)
                self._index.append(None)
                base = sum(len(x) for x in lengths)
                recs = list(parse_sequences(path))
                for k, rec in enumerate(recs):
                    self._parsed[base + k] = rec
                file_of.append(np.full(len(recs), fi, dtype=np.int64))
                local_of.append(np.arange(len(recs), dtype=np.int64))
                lengths.append(np.array([len(r.seq) for r in recs], dtype=np.int64))
                continue
            idx = _lib.fasta_index(buf)
            self._maps.append(mm)
            self._bufs.append(buf)
            self._index.append(idx)
            file_of.append(np.full(idx[0].size, fi, dtype=np.int64))
            local_of.append(np.arange(idx[0].size, dtype=np.int64))
            lengths.append(idx[4])
        self._file_of = np.concatenate(file_of) if file_of else np.zeros(0, dtype=np.int64)
        self._local_of = np.concatenate(local_of) if local_of else np.zeros(0, dtype=np.int64)
        self.lengths = np.concatenate(lengths) if lengths else np.zeros(0, dtype=np.int64)
        self.ids = _Headers(self, True)
        self.headers = _Headers(self, False)

    def __len__(self):
        return len(self.lengths)

    def _header(self, i):
        rec = self._parsed.get(i)
        if rec is not None:
            return rec.description
        fi, k = int(self._file_of[i]), int(self._local_of[i])
        off, n = int(self._index[fi][0][k]), int(self._index[fi][1][k])
        return self._maps[fi][off:off + n].decode("utf-8", "replace")

    def _read(self, lo, hi):
        out = []
        for i in range(lo, hi):
            rec = self._parsed.get(i)
            if rec is None:
                fi, k = int(self._file_of[i]), int(self._local_of[i])
                idx = self._index[fi]
                lines = self._maps[fi][int(idx[2][k]):int(idx[3][k])].split(b"\n")
                rec = _record(self._header(i), [ln.strip().decode("latin-1") for ln in lines])
            out.append(rec)
        return out

    def _bulk_strings(self, want_id):
        """every id (or header) as one list, when every file is plain and its headers ASCII; else None"""
        key = "ids" if want_id else "headers"
        if key not in self._bulk:
            out = []
            for fi in range(len(self.files)):
                n = 0 if self._index[fi] is None else int(self._index[fi][0].size)
                tables = self._span_tables_of(fi)
                if tables is None:
                    out = None
                    break
                if n:
                    from . import _lib
                    blob = _lib.gather_spans(self._bufs[fi], tables[0 if want_id else 1])
                    out.extend(blob.decode("ascii").split("\n")[:n])
            self._bulk[key] = out
        return self._bulk[key]

    def _span_tables_of(self, fi):
        from . import _lib
        if self._index[fi] is None:
            return None
        if fi not in self._spans:
            idx = self._index[fi]
            ids, ascii_ = _lib.fasta_ids(self._bufs[fi], idx[0], idx[1])
            self._spans[fi] = (ids, np.stack([idx[0], idx[1]], axis=1)) if ascii_ else None
        return self._spans[fi]

    def _span_tables(self, lo, hi):
        from . import _lib
        if hi <= lo or (self._parsed and any(i in self._parsed for i in range(lo, hi))):
            return None
        fi = int(self._file_of[lo])
        if int(self._file_of[hi - 1]) != fi:
            return None
        tables = self._span_tables_of(fi)
        if tables is None:
            return None
        a, b = int(self._local_of[lo]), int(self._local_of[hi - 1]) + 1
        return self._bufs[fi], tables[0][a:b], tables[1][a:b]

    def _pack(self, lo, hi, lut):
        from . import _lib
        if hi <= lo:
            return np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        if self._parsed and any(i in self._parsed for i in range(lo, hi)):
            return None
        codes, offsets, base, i = [], [], 0, lo
        while i < hi:                          # one native call per run of records that sit in the same file
            fi = int(self._file_of[i])
            j = i
            while j < hi and int(self._file_of[j]) == fi:
                j += 1
            idx = self._index[fi]
            a, b = int(self._local_of[i]), int(self._local_of[j - 1]) + 1
            c, o = _lib.fasta_encode(self._bufs[fi], idx[2], idx[3], idx[4], a, b, lut, pack.SEP)
            codes.append(c)
            offsets.append(o + base)
            base += c.size
            i = j
        lengths = np.array(self.lengths[lo:hi], dtype=np.int64)
        if len(codes) == 1:
            return codes[0], offsets[0], lengths
        return np.concatenate(codes), np.concatenate(offsets), lengths

    def __getitem__(self, key):
        if isinstance(key, slice):
            lo, hi, step = key.indices(len(self))
            if step != 1:
                raise ValueError("LazyFasta slices are contiguous")
            return FastaSlice(self, lo, max(lo, hi))
        if key < 0:
            key += len(self)
        return self._read(key, key + 1)[0]

    def __iter__(self):
        for lo in range(0, len(self), 1024):
            for rec in self._read(lo, min(len(self), lo + 1024)):
                yield rec


_LAZY_CACHE = {}


def open_lazy(fasta_files):
    """LazyFasta(fasta_files), indexed once per process: the background pass (rnascan.py:440-465) and the scan
    (rnascan.py:379) walk the same files.  Keyed by path, size and mtime of every file; the two most recent are kept (the
    sequence and the structure FASTA of a two-file run)."""
    files = [fasta_files] if isinstance(fasta_files, str) else list(fasta_files)
    try:
        key = tuple((f, os.path.getsize(f), os.stat(f).st_mtime_ns) for f in files)
    except OSError:
        return LazyFasta(files)
    got = _LAZY_CACHE.pop(key, None)
    if got is None:
        got = LazyFasta(files)
        while len(_LAZY_CACHE) >= 2:
            _LAZY_CACHE.pop(next(iter(_LAZY_CACHE)))
    _LAZY_CACHE[key] = got                               # most recently used last
    return got


def _record(header, chunks):
    words = header.split(None, 1)
    rid = words[0] if words else ""
    return Record(rid, header, "".join(chunks).replace(" ", ""))


def preprocess_seq(seq, target_is_rna, source_is_rna=False):
    """rnascan.py:186-197: transcribe (T->U, t->u) and upper-case only when the
    target alphabet is RNA and the source is not declared RNA; otherwise the
    sequence is returned untouched (structure strings are not upper-cased)."""
    if target_is_rna and not source_is_rna:
        return seq.replace("T", "U").replace("t", "u").upper()
    return seq


def _count_letters_natively(fasta_files, lut, positions=1 << 26):
    """counts of the codes 0..7 of plain FASTA files through the native packer (batches of ``positions`` letters), or
    None when a file cannot be mapped (compressed, CR-only line ends)"""
    files = [fasta_files] if isinstance(fasta_files, str) else list(fasta_files)
    if any(os.path.splitext(f)[1] in (".gz", ".bz2") for f in files):
        return None
    from . import _lib, shard
    lazy = open_lazy(files)
    counts = np.zeros(256, dtype=np.int64)
    for lo, hi in shard.batches(lazy.lengths, 0, len(lazy), positions):
        if hi > lo:
            packed = lazy[lo:hi].pack_letters(lut)
            if packed is None:
                return None
            counts += _lib.count_bytes(packed[0])
    return counts[:8]


def _count_rna_natively(fasta_files, positions=1 << 26):
    """A, C, G, U counts (preprocess_seq + count, natively)"""
    counted = _count_letters_natively(fasta_files, pack._RNA_LUT, positions)
    return None if counted is None else counted[:4]


def compute_background(fasta_files, letters, verbose=True):
    """rnascan.py:440-465: letter counts over all (preprocessed) records with a
    +1 pseudocount per alphabet letter."""
    eprint("Calculating background probabilities...")
    content = {}
    total = len(letters)
    is_rna = is_rna_letters(letters)
    if is_rna:
        counted, where = _count_rna_natively(fasta_files), pack.RNA_LETTERS
    else:
        # a generic alphabet is counted as written (no upper-casing, rnascan.py:186-197; Seq.count is case-sensitive):
        # with the case-keeping LUT the codes 0..6 are exactly the upper-case letters
        counted, where = _count_letters_natively(fasta_files, pack.letter_lut(letters, keep_case=True)), letters
    if counted is not None:                    # the same counts from the packed codes (preprocess_seq + count, natively)
        for letter in letters:
            content[letter] = int(counted[where.index(letter)])
            total += content[letter]
    else:
        for rec in parse_sequences(fasta_files):
            s = preprocess_seq(rec.seq, is_rna)
            for letter in letters:
                amount = s.count(letter)
                content[letter] = content.get(letter, 0) + amount
                total += amount
    pct_sum = 0.0
    for letter, count in content.items():
        content[letter] = (float(count) + 1) / total
        if content[letter] <= 0.05:
            warnings.warn("Letter %s has low content: %0.2f" % (letter, content[letter]), Warning)
        pct_sum += content[letter]
    if verbose:
        eprint(dict(content))
    assert abs(1.0 - pct_sum) < 0.0001, "Background sums to %f" % pct_sum
    return content


def load_background(bg_file, uniform, fasta_files, letters, verbose=True):
    """rnascan.py:468-484: custom dict-literal file, else computed, else None (= uniform)."""
    if bg_file:
        eprint("Reading custom background probabilities from %s" % bg_file)
        with open(bg_file, "r") as fin:
            bg = ast.literal_eval(fin.read())
            eprint(dict(bg))
        return bg
    if not uniform:
        return compute_background(fasta_files, letters, verbose)
    return None


def read_profile(struct_file):
    """An averaged-structure profile file (written by pfmutil.py:61-87): header
    ``PO`` + letters, one row per position.  Returns (letters, float64 [L][n]).

    Parsed the way rnascan.py:296-297 does it (``pd.read_table`` then ``del struct['PO']``),
    so the float64 values are the ones the reference computes with, bit for bit (pandas'
    default converter is not always the correctly rounded one ``float()`` is).  Plain files go through
    ``pfmscan_profile_parse``, the same converter restated natively (~10x faster per file, and it releases the GIL so
    that ``read_profiles`` parses a batch of files on all cores); anything unusual is left to pandas itself."""
    from . import _lib
    with open(struct_file, "rb") as fh:
        data = fh.read()
    head = data.split(b"\n", 1)[0].rstrip(b"\r").split(b"\t")
    native = os.environ.get("RNASCAN_PROFILE_PARSER", "native") != "pandas"       # "pandas": A/B of the two parsers
    # the native parser drops the FIRST column; the reference deletes the column NAMED 'PO' (rnascan.py:297), so only a
    # file whose first column is 'PO' may take it -- any other header is pandas' to judge (below: by name, as the reference)
    if native and len(head) >= 2 and head[0] == b"PO" and len(set(head)) == len(head) and \
            all(h and h.strip() == h and b'"' not in h for h in head):
        prof = _lib.profile_parse(data, len(head) - 1)
        if prof is not None and prof.shape[0] > 0:
            return [h.decode("utf-8") for h in head[1:]], prof
    import io
    import pandas as pd
    df = pd.read_table(io.BytesIO(data))
    if "PO" in df.columns:
        del df["PO"]
    else:
        df = df.iloc[:, 1:]
    letters = [str(c) for c in df.columns]
    prof = np.ascontiguousarray(df.to_numpy(dtype=np.float64)).reshape(-1, len(letters))
    return letters, prof


def read_profiles(paths, threads=None):
    """read_profile for a batch of files on a thread pool (the native parser runs without the GIL)"""
    paths = list(paths)
    if len(paths) < 4:
        return [read_profile(p) for p in paths]
    from concurrent.futures import ThreadPoolExecutor
    if threads is None:
        threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        return list(pool.map(read_profile, paths))


def list_profiles(directory):
    """rnascan.py:351 + :370-372: structure.<id>.txt files of a directory and the
    Sequence_ID recovered from each name."""
    files = glob.glob(directory + "/structure.*.txt")
    out = []
    for f in files:
        m = re.search(r"^structure\.(.*)\.txt$", os.path.basename(f))
        out.append((m.group(1), f))
    return out
