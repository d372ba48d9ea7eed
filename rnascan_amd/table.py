"""Streaming hit-table writer.

The reference builds one DataFrame per record, concatenates them, merges the sequence
and structure tables and prints with ``DataFrame.to_csv(sep='\\t', index=False)``
(rnascan.py:284-286, :407-408, :422-423, :559-567).  At permissive thresholds the
table has 10^8+ rows and the DataFrames dwarf the scan.  ``TsvWriter`` writes the same
bytes -- same column order, ``Match_ID`` appended last and numbered 1..n across chunks
(rnascan.py:329-332), float32 scores as the shortest float32 repr (``14.259``), float64
scores as ``repr(float)``, NaN as the empty field -- from column arrays, one chunk (one
shard, one batch) at a time, without ever holding the whole table.  The rows are formatted
by ``pfmscan_tsv_format`` (native, parallel); the scanners hand over their hit columns in
the compact forms below, so no per-row Python object is ever made on the streaming path.
"""
import numpy as np

from . import _lib


class Indexed(object):
    """A string column given as (values, index): row r holds ``values[index[r]]`` (record ids, descriptions,
    motif ids -- one Python string per distinct value instead of one per row)."""

    def __init__(self, values, index):
        self.values = values
        self.index = np.asarray(index, dtype=np.int64)

    def __len__(self):
        return self.index.shape[0]

    def materialize(self):
        used, inv = np.unique(self.index, return_inverse=True)
        vals = np.empty(used.size, dtype=object)
        vals[:] = [self.values[i] for i in used.tolist()]
        return vals[inv]


class Spans(object):
    """A string column given as byte spans of a buffer: row r holds ``buffer[off:off + n]`` for (off, n) =
    ``spans[index[r]]`` -- ids and headers straight from the mapped FASTA file (ASCII), never made into Python strings
    on the streaming path; csv quoting is applied by the native formatter."""

    def __init__(self, buffer, spans, index):
        self.buffer = buffer
        self.spans = np.ascontiguousarray(spans, dtype=np.int64)
        self.index = np.asarray(index, dtype=np.int64)

    def __len__(self):
        return self.index.shape[0]

    def materialize(self):
        used, inv = np.unique(self.index, return_inverse=True)
        vals = np.empty(used.size, dtype=object)
        vals[:] = [self.buffer[o:o + n].tobytes().decode("ascii") for o, n in self.spans[used].tolist()]
        return vals[inv]


class Windows(object):
    """The ``Sequence`` column given as stream positions into a code array: row r holds the ``m`` letters
    ``letters[codes[pos[r] + j]]`` (rnascan.py:272 slices the record string per hit)."""

    def __init__(self, codes, pos, m, letters, cased=False):
        self.codes = codes
        self.pos = np.asarray(pos, dtype=np.int64)
        self.m = int(m)
        self.letters = letters
        self.cased = cased                      # codes 8..15 = the same letters written in lower case (pack.CASE_BIT)

    def __len__(self):
        return self.pos.shape[0]

    def blob(self):
        """the 16 letters of codes 0..15 (PFMSCAN_TSV_WINDOW)"""
        upper = self.letters.ljust(8, "?")
        return (upper + (self.letters.lower().ljust(8, "?") if self.cased else "?" * 8)).encode("ascii")

    def materialize(self):
        lut = np.frombuffer(self.blob(), dtype=np.uint8)
        if self.pos.size == 0 or self.m == 0:
            return np.full(self.pos.size, "", dtype=object)
        win = lut[self.codes[self.pos[:, None] + np.arange(self.m)] & 15]
        return np.ascontiguousarray(win).view("S%d" % self.m).reshape(-1).astype(str).astype(object)


def column_length(col):
    if isinstance(col, (Indexed, Windows, Spans)):
        return len(col)
    if isinstance(col, str) or np.ndim(col) == 0:
        return None
    return len(col)


def _rows(col, a, b):
    """rows [a, b) of a column, same kind"""
    if isinstance(col, Indexed):
        return Indexed(col.values, col.index[a:b])
    if isinstance(col, Spans):
        return Spans(col.buffer, col.spans, col.index[a:b])
    if isinstance(col, Windows):
        return Windows(col.codes, col.pos[a:b], col.m, col.letters, col.cased)
    if column_length(col) is None:
        return col
    return col[a:b]


def to_frame(columns, order=None):
    """compact columns -> pandas DataFrame (the non-streaming consumers: joins, multi-rank gathers, the API)"""
    import pandas as pd
    order = list(order or columns.keys())
    n = next((k for k in (column_length(columns[c]) for c in order) if k is not None), 0)
    data = {}
    for c in order:
        col = columns[c]
        if isinstance(col, (Indexed, Windows, Spans)):
            col = col.materialize()
        elif column_length(col) is None:
            col = np.full(n, col, dtype=object if isinstance(col, str) else None)
        data[c] = col
    return pd.DataFrame(data, columns=order)


def _quote(field):
    """csv.QUOTE_MINIMAL as ``to_csv(sep='\\t')`` applies it (rnascan.py:559-567): a string field holding the
    delimiter, the quote character or a line break is wrapped in double quotes, embedded quotes are doubled.
    ``Description`` is the whole FASTA header, so this is reachable from ordinary input."""
    if "\t" in field or '"' in field or "\n" in field or "\r" in field:
        return '"' + field.replace('"', '""') + '"'
    return field


def _blob(strings):
    """list of field strings -> (offsets int64 [n + 1], utf-8 bytes)"""
    enc = [_quote(s).encode("utf-8") for s in strings]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        np.cumsum([len(e) for e in enc], out=off[1:])
    return off, b"".join(enc)


def _descriptor(col, n):
    """one column -> (kind, data, aux, blob, width) of pfmscan_tsv_format, formatted the way pandas' to_csv does"""
    if isinstance(col, str):
        b = _quote(col).encode("utf-8")
        return (_lib.TSV_CONST, b, None, None, len(b))
    if isinstance(col, Indexed):
        if len(col) != n:
            raise ValueError("column length mismatch")
        if isinstance(col.values, (list, tuple)) and len(col.values) <= max(4096, n // 8):
            off, blob = _blob(col.values)            # a short list (motif ids): every value, the index as it is
            return (_lib.TSV_INDEXED, col.index, off, blob, 0)
        # many (or lazily decoded) values: only those some row uses are stringified
        used, inv = np.unique(col.index, return_inverse=True)
        off, blob = _blob([col.values[i] for i in used.tolist()])
        return (_lib.TSV_INDEXED, inv.astype(np.int64), off, blob, 0)
    if isinstance(col, Spans):
        if len(col) != n:
            raise ValueError("column length mismatch")
        return (_lib.TSV_SPAN, col.index, col.spans, col.buffer, 0)
    if isinstance(col, Windows):
        if len(col) != n:
            raise ValueError("column length mismatch")
        return (_lib.TSV_WINDOW, col.pos, col.codes, col.blob(), col.m)
    if isinstance(col, (list, tuple)):
        if len(col) != n:
            raise ValueError("column length mismatch")
        if col and not isinstance(col[0], str):
            return _descriptor(np.asarray(col), n)
        off, blob = _blob(col)
        return (_lib.TSV_INDEXED, np.arange(n, dtype=np.int64), off, blob, 0)
    a = np.asarray(col)
    if a.ndim == 0:
        return _descriptor(a.reshape(1).repeat(n), n) if a.dtype.kind in "fiu" else _descriptor(str(a.item()), n)
    if a.shape[0] != n:
        raise ValueError("column length mismatch")
    if a.dtype == np.float32:
        return (_lib.TSV_F32, a, None, None, 0)
    if a.dtype.kind == "f":
        return (_lib.TSV_F64, a.astype(np.float64, copy=False), None, None, 0)
    if a.dtype.kind in "iu":
        return (_lib.TSV_I64, a.astype(np.int64, copy=False), None, None, 0)
    if a.dtype.kind == "S":
        return (_lib.TSV_FIXED, a, None, None, a.dtype.itemsize)
    strings = [x if isinstance(x, str) else ("" if x is None or (isinstance(x, float) and x != x) else str(x)) for x in a.tolist()]
    off, blob = _blob(strings)
    return (_lib.TSV_INDEXED, np.arange(n, dtype=np.int64), off, blob, 0)


class TsvWriter(object):
    """write_chunk(columns) appends rows; columns is an ordered mapping name -> array | list | scalar |
    Indexed | Spans | Windows (a DataFrame's columns work as arrays)."""

    MAX_ROWS = 1 << 22                          # rows per native call (~0.5 GB of row buffer at the widest tables)
    ASYNC_ROWS = 1 << 18                        # chunks of at least this many rows are written out beside the next one's formatting

    def __init__(self, out, columns, match_id=True, header=True):
        self.out = out
        self.columns = list(columns)
        self.match_id = match_id
        self.rows = 0
        # two row buffers, reused from chunk to chunk: while one is being written out by the writer thread the next
        # chunk is formatted into the other (a table of 10^7+ rows is bound by exactly these two steps)
        self._scratch = [[None], [None]]
        self._busy = [None, None]                # the write still reading each buffer
        self._turn = 0
        self._pool = None
        if header:
            out.write("\t".join([_quote(c) for c in self.columns] + (["Match_ID"] if match_id else [])) + "\n")

    def write_chunk(self, data, n=None):
        if n is None:
            n = next((k for k in (column_length(data[c]) for c in self.columns) if k is not None), 0)
        if n == 0:
            return
        if n > self.MAX_ROWS:                    # an all-scores table (-m ' -inf'): bound the row buffer, not the table
            for a in range(0, n, self.MAX_ROWS):
                b = min(n, a + self.MAX_ROWS)
                self.write_chunk({c: _rows(data[c], a, b) for c in self.columns}, b - a)
            return
        desc = [_descriptor(data[c], n) for c in self.columns]
        k = self._turn
        self._turn ^= 1
        if self._busy[k] is not None:
            self._busy[k].result()               # the write that was reading this buffer (raises what it raised)
            self._busy[k] = None
        pieces = _lib.tsv_format(desc, n, self.rows + 1 if self.match_id else -1, scratch=self._scratch[k])
        raw = getattr(self.out, "buffer", None)
        if raw is not None and getattr(self.out, "encoding", "utf-8").lower().replace("-", "") == "utf8":
            # a text stream over a byte stream: no decode / re-encode of the rows.  Large chunks go out on the writer
            # thread (ONE thread: the chunks stay in order) while the next one is formatted
            if n >= self.ASYNC_ROWS:
                if self._pool is None:
                    from concurrent.futures import ThreadPoolExecutor
                    self._pool = ThreadPoolExecutor(max_workers=1)
                    self.out.flush()
                self._busy[k] = self._pool.submit(self._write_pieces, raw, pieces)
            else:
                self.finish()
                self.out.flush()
                self._write_pieces(raw, pieces)
        else:
            self.out.write(b"".join(pieces).decode("utf-8"))
        self.rows += n

    @staticmethod
    def _write_pieces(raw, pieces):
        for piece in pieces:
            raw.write(piece)

    def finish(self):
        """wait for the writer thread (call before anything else writes to the stream, and at the end)"""
        for k in (self._turn, self._turn ^ 1):   # oldest first
            if self._busy[k] is not None:
                self._busy[k].result()
                self._busy[k] = None

    def close(self):
        self.finish()
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_frame(out, df, match_id=True, chunk=1 << 20):
    """stream a DataFrame through the writer (same bytes as to_csv after _add_match_id)"""
    w = TsvWriter(out, list(df.columns), match_id)
    for lo in range(0, len(df), chunk):
        part = df.iloc[lo:lo + chunk]
        w.write_chunk({c: part[c].to_numpy() for c in df.columns}, len(part))
    w.close()
    return w.rows
