"""Packed averaged-structure profile store.

The reference keeps one ``structure.<seq_id>.txt`` TSV per record
(scripts/run_folding:66, format of pfmutil.py:61-87) and parses each of them with
pandas inside the scan loop (rnascan.py:296-297), found by an unsorted glob
(:351).  At 100k records the text parsing dominates everything else.  This module
converts such a directory ONCE into the exact layout the kernels read --

    <store>/profile.f32 | profile.f64   [n_pos][7] row-major, one zero row after each record
    <store>/index.json                  ids, lengths, column letters, dtype, n_pos

-- which is then memory-mapped and handed to the device as is (no per-record copy).
``rnascan`` accepts a store wherever it accepts an averaged-structure directory.
"""
import json
import os

import numpy as np

from . import fasta, pack

INDEX = "index.json"
FORMAT_VERSION = 1


def is_store(path):
    return os.path.isdir(path) and os.path.exists(os.path.join(path, INDEX))


def build_store(directory, store_dir, dtype=np.float64):
    """Convert ``directory/structure.*.txt`` into a packed store.  Records are stored
    sorted by Sequence_ID (the glob order of the reference is filesystem dependent);
    returns the number of records."""
    files = sorted(fasta.list_profiles(directory))
    if not files:
        raise IOError("No averaged structure files found")
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("dtype must be float32 or float64")
    os.makedirs(store_dir, exist_ok=True)
    ids, lengths, letters0 = [], [], None
    name = "profile.f32" if dtype == np.float32 else "profile.f64"
    tmp = os.path.join(store_dir, name + ".tmp")
    sep = np.zeros((1, 7), dtype=dtype)
    with open(tmp, "wb") as out:
        for lo in range(0, len(files), 512):                     # 512 files at a time parsed on all cores, written in order
            group = files[lo:lo + 512]
            for (sid, path), (letters, prof) in zip(group, fasta.read_profiles([p for _, p in group])):
                if letters0 is None:
                    letters0 = list(letters)
                elif list(letters) != letters0:
                    raise ValueError("%s: column order %s differs from %s" % (path, letters, letters0))
                if prof.shape[1] != 7:
                    raise ValueError("%s: expected 7 structure columns" % path)
                out.write(np.ascontiguousarray(prof, dtype=dtype).tobytes())
                out.write(sep.tobytes())
                ids.append(sid)
                lengths.append(int(prof.shape[0]))
    os.replace(tmp, os.path.join(store_dir, name))
    index = {"format": FORMAT_VERSION, "dtype": dtype.name, "letters": letters0, "ids": ids, "lengths": lengths,
             "n_pos": int(sum(lengths) + len(lengths)), "file": name, "separator_rows": "one zero row after each record"}
    with open(os.path.join(store_dir, INDEX), "w") as f:
        json.dump(index, f)
    return len(ids)


class ProfileStore(object):
    """Memory-mapped packed profiles; ``stream()`` is what the engine scans."""

    def __init__(self, store_dir):
        with open(os.path.join(store_dir, INDEX)) as f:
            idx = json.load(f)
        if idx.get("format") != FORMAT_VERSION:
            raise ValueError("unknown profile store format %r" % idx.get("format"))
        self.ids = idx["ids"]
        self.letters = idx["letters"]
        self.lengths = np.asarray(idx["lengths"], dtype=np.int64)
        self.dtype = np.dtype(idx["dtype"])
        self.n_pos = int(idx["n_pos"])
        with open(os.path.join(store_dir, idx["file"]), "rb") as fh:
            st = os.fstat(fh.fileno())
            self.profile = np.memmap(fh, dtype=self.dtype, mode="r", shape=(self.n_pos, 7))
        # WHICH file is mapped (the path may come to name another one: a store re-packed by rename while this one is open);
        # the staged uploader reads the file beside the mapping only while the path still names this file (_lib._register_mapping)
        self.profile._mapped_file_id = (st.st_dev, st.st_ino, st.st_size)
        self.offsets = np.zeros(len(self.lengths), dtype=np.int64)
        if len(self.lengths) > 1:
            self.offsets[1:] = np.cumsum(self.lengths[:-1] + 1)

    def stream(self, lo=0, hi=None):
        """The packed stream of records [lo, hi) -- a VIEW of the mapped file."""
        hi = len(self.ids) if hi is None else hi
        if hi <= lo:
            return pack.Stream(None, np.zeros((0, 7), self.dtype), np.zeros(0, np.int64), np.zeros(0, np.int64))
        a = int(self.offsets[lo])
        b = int(self.offsets[hi - 1] + self.lengths[hi - 1] + 1)
        return pack.Stream(None, self.profile[a:b], self.offsets[lo:hi] - a, self.lengths[lo:hi])

    def named(self, lo=0, hi=None):
        """records [lo, hi) as (Sequence_ID, letters, [L][7]) views, the scanner's input form"""
        hi = len(self.ids) if hi is None else hi
        return [(self.ids[r], self.letters, self.profile[int(self.offsets[r]):int(self.offsets[r] + self.lengths[r])])
                for r in range(lo, hi)]


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="rnascan-pack-profiles",
                                 description="Pack a directory of structure.<id>.txt files into a binary profile store")
    ap.add_argument("directory")
    ap.add_argument("store")
    ap.add_argument("--dtype", choices=["float64", "float32"], default="float64")
    args = ap.parse_args(argv)
    n = build_store(args.directory, args.store, np.dtype(args.dtype))
    fasta.eprint("Packed %d profiles into %s" % (n, args.store))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
