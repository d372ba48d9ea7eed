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


class LazyFasta(object):
    """A FASTA file as a sliceable sequence of Records that holds only an INDEX in memory: id, header, byte offset and
    letter count per record, from one cheap pass over the bytes.  ``lazy[a:b]`` seeks to record a and parses b - a
    records, so a rank of a sharded run (shard.scan_sharded) reads only its own share and a batch only its own
    records -- the reference hands every record to a pool worker through one iterator (rnascan.py:379-395).
    Compressed input (.gz / .bz2) cannot be seeked: its records are parsed once and kept."""

    def __init__(self, fasta_files):
        self.files = [fasta_files] if isinstance(fasta_files, str) else list(fasta_files)
        self.ids, self.headers, self.lengths = [], [], []
        self._where = []                       # (file index, byte offset) per record, or a parsed Record
        for fi, path in enumerate(self.files):
            if os.path.splitext(path)[1] in (".gz", ".bz2"):
                for rec in parse_sequences(path):
                    self._add(rec.id, rec.description, len(rec.seq), rec)
                continue
            with open(path, "rb") as fh:
                off, cur, n = 0, None, 0
                for line in fh:
                    if line.startswith(b">"):
                        if cur is not None:
                            self._add(cur[0], cur[1], n, (fi, cur[2]))
                        header = line[1:].rstrip(b"\r\n").decode("utf-8", "replace")
                        words = header.split(None, 1)
                        cur, n = (words[0] if words else "", header, off), 0
                    elif cur is not None:
                        n += len(line.strip().replace(b" ", b""))
                    off += len(line)
                if cur is not None:
                    self._add(cur[0], cur[1], n, (fi, cur[2]))

    def _add(self, rid, header, n, where):
        self.ids.append(rid)
        self.headers.append(header)
        self.lengths.append(n)
        self._where.append(where)

    def __len__(self):
        return len(self.ids)

    def _read(self, lo, hi):
        out = []
        i = lo
        while i < hi:
            w = self._where[i]
            if isinstance(w, Record):
                out.append(w)
                i += 1
                continue
            # the run of records lo.. that sit in the same file: one seek, parse until the run ends
            fi, off = w
            j = i
            while j < hi and not isinstance(self._where[j], Record) and self._where[j][0] == fi:
                j += 1
            with open(self.files[fi], "rb") as fh:
                fh.seek(off)
                header, chunks, got = None, [], 0
                for line in fh:
                    if line.startswith(b">"):
                        if header is not None:
                            out.append(_record(header, chunks))
                            got += 1
                            if got == j - i:
                                header = None
                                break
                        header, chunks = line[1:].rstrip(b"\r\n").decode("utf-8", "replace"), []
                    elif header is not None:
                        chunks.append(line.strip().decode("latin-1"))
                if header is not None:
                    out.append(_record(header, chunks))
            i = j
        return out

    def __getitem__(self, key):
        if isinstance(key, slice):
            lo, hi, step = key.indices(len(self))
            if step != 1:
                raise ValueError("LazyFasta slices are contiguous")
            return self._read(lo, max(lo, hi))
        if key < 0:
            key += len(self)
        return self._read(key, key + 1)[0]

    def __iter__(self):
        for lo in range(0, len(self), 1024):
            for rec in self._read(lo, min(len(self), lo + 1024)):
                yield rec


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


def compute_background(fasta_files, letters, verbose=True):
    """rnascan.py:440-465: letter counts over all (preprocessed) records with a
    +1 pseudocount per alphabet letter."""
    eprint("Calculating background probabilities...")
    content = {}
    total = len(letters)
    is_rna = is_rna_letters(letters)
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
    default converter is not always the correctly rounded one ``float()`` is)."""
    import pandas as pd
    df = pd.read_table(struct_file)
    if "PO" in df.columns:
        del df["PO"]
    else:
        df = df.iloc[:, 1:]
    letters = [str(c) for c in df.columns]
    prof = np.ascontiguousarray(df.to_numpy(dtype=np.float64)).reshape(-1, len(letters))
    return letters, prof


def list_profiles(directory):
    """rnascan.py:351 + :370-372: structure.<id>.txt files of a directory and the
    Sequence_ID recovered from each name."""
    files = glob.glob(directory + "/structure.*.txt")
    out = []
    for f in files:
        m = re.search(r"^structure\.(.*)\.txt$", os.path.basename(f))
        out.append((m.group(1), f))
    return out
