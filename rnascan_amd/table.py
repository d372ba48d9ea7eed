"""Streaming hit-table writer.

The reference builds one DataFrame per record, concatenates them, merges the sequence
and structure tables and prints with ``DataFrame.to_csv(sep='\\t', index=False)``
(rnascan.py:284-286, :407-408, :422-423, :559-567).  At permissive thresholds the
table has 10^8+ rows and the DataFrames dwarf the scan.  ``TsvWriter`` writes the same
bytes -- same column order, ``Match_ID`` appended last and numbered 1..n across chunks
(rnascan.py:329-332), float32 scores as the shortest float32 repr (``14.259``), float64
scores as ``repr(float)``, NaN as the empty field -- from column arrays, one chunk (one
shard, one batch) at a time, without ever holding the whole table.
"""
import numpy as np


def _quote(field):
    """csv.QUOTE_MINIMAL as ``to_csv(sep='\\t')`` applies it (rnascan.py:559-567): a string field holding the
    delimiter, the quote character or a line break is wrapped in double quotes, embedded quotes are doubled.
    ``Description`` is the whole FASTA header, so this is reachable from ordinary input."""
    if "\t" in field or '"' in field or "\n" in field or "\r" in field:
        return '"' + field.replace('"', '""') + '"'
    return field


def _strings(col, n):
    """column -> list of n field strings, formatted the way pandas' to_csv does"""
    if isinstance(col, str):
        return [_quote(col)] * n
    if isinstance(col, (list, tuple)):
        if len(col) != n:
            raise ValueError("column length mismatch")
        if col and not isinstance(col[0], str):
            return _strings(np.asarray(col), n)
        return [_quote(x) for x in col]
    a = np.asarray(col)
    if a.ndim == 0:
        return [_strings(a.reshape(1), 1)[0]] * n
    if a.shape[0] != n:
        raise ValueError("column length mismatch")
    if a.dtype.kind == "f":
        s = a.astype(str)                       # shortest repr of the column's own precision
        if np.isnan(a).any():
            s = s.astype(object)
            s[np.isnan(a)] = ""                 # na_rep=''
        return s.tolist()
    if a.dtype.kind in "iu":
        return a.astype(str).tolist()
    return [_quote(x) if isinstance(x, str) else str(x) for x in a.tolist()]


class TsvWriter(object):
    """write_chunk(columns) appends rows; columns is an ordered mapping name -> array | list | scalar."""

    def __init__(self, out, columns, match_id=True):
        self.out = out
        self.columns = list(columns)
        self.match_id = match_id
        self.rows = 0
        out.write("\t".join([_quote(c) for c in self.columns] + (["Match_ID"] if match_id else [])) + "\n")

    def write_chunk(self, data, n=None):
        if n is None:
            n = len(next(v for v in data.values() if not isinstance(v, str) and np.ndim(v) > 0))
        if n == 0:
            return
        cols = [_strings(data[c], n) for c in self.columns]
        if self.match_id:
            cols.append(np.arange(self.rows + 1, self.rows + n + 1).astype(str).tolist())
        self.out.write("\n".join("\t".join(t) for t in zip(*cols)))
        self.out.write("\n")
        self.rows += n


def write_frame(out, df, match_id=True, chunk=1 << 20):
    """stream a DataFrame through the writer (same bytes as to_csv after _add_match_id)"""
    w = TsvWriter(out, list(df.columns), match_id)
    for lo in range(0, len(df), chunk):
        part = df.iloc[lo:lo + chunk]
        w.write_chunk({c: part[c].to_numpy() for c in df.columns}, len(part))
    return w.rows
