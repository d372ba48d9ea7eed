"""Native host ingest / output (pfmscan_fasta_index, pfmscan_fasta_encode, pfmscan_tsv_format; no device needed)
against the Python statements of the same behaviour: parse_sequences + preprocess_seq + encode_rna
(rnascan.py:170-174, :177-204, _pwm.c:41-63) and DataFrame.to_csv(sep='\\t', index=False) (rnascan.py:555-567)."""
import io
import os

import numpy as np
import pandas as pd
import pytest
from hypothesis import given, settings, strategies as st

from rnascan_amd import _lib, fasta, pack, table

NASTY = (b"junk before the first header\nACGT\n"
         b">r1 first record\r\nACGTacgu\r\n  AC GU \t\n\n"
         b">r2\n"
         b">r3 only\x0bodd white\tspace\nNNNN\x0b\nAC\tGU\n"
         b">\nAC>GU\n"
         b">r5 caf\xc3\xa9 \"quoted\"\tTAB\n" + b"ACGU" * 50 + b"\n" + b"acgt" * 50 + b"\n"
         b">last no newline\nGGGG")


def _reference_pack(path):
    recs = list(fasta.parse_sequences(path))
    codes = [pack.encode_rna(fasta.preprocess_seq(r.seq, True)) for r in recs]
    return recs, pack.pack(codes)


def _write(tmp_path, name, data):
    p = os.path.join(str(tmp_path), name)
    with open(p, "wb") as f:
        f.write(data)
    return p


@pytest.mark.parametrize("data", [NASTY, b"", b"no header at all\nACGU\n", b">only header", b">a\nAC\n>b\nGU\n", b"\n\n>x\n\n\nA\n\n"])
def test_native_fasta_matches_the_python_parser(tmp_path, data):
    path = _write(tmp_path, "a.fa", data)
    recs, want = _reference_pack(path) if data else ([], None)
    lazy = fasta.LazyFasta(path)
    assert len(lazy) == len(recs)
    assert list(lazy.ids) == [r.id for r in recs]
    assert list(lazy.headers) == [r.description for r in recs]
    assert list(lazy.lengths) == [len(r.seq) for r in recs]
    assert [tuple(r) for r in lazy] == [tuple(r) for r in recs]
    if not recs:
        return
    codes, offsets, lengths = lazy[0:len(lazy)].pack_rna()
    assert np.array_equal(codes, want.codes) and np.array_equal(offsets, want.offsets) and np.array_equal(lengths, want.lengths)
    for lo in range(len(recs)):                       # every sub-range, every thread count
        for hi in range(lo, len(recs) + 1):
            sub = lazy[lo:hi]
            assert [tuple(r) for r in sub] == [tuple(r) for r in recs[lo:hi]]
            assert list(sub.ids) == [r.id for r in recs[lo:hi]]
            c, o, ln = sub.pack_rna()
            ref = pack.pack([pack.encode_rna(fasta.preprocess_seq(r.seq, True)) for r in recs[lo:hi]]) if hi > lo else None
            if ref is None:
                assert c.size == 0 and o.size == 0
            else:
                assert np.array_equal(c, ref.codes) and np.array_equal(o, ref.offsets) and np.array_equal(ln, ref.lengths)


line = st.one_of(
    st.text(alphabet="ACGUTacgutNn", min_size=0, max_size=30),
    st.text(alphabet="ACGU \t>x", min_size=0, max_size=12),
    st.builds(lambda s: ">" + s, st.text(alphabet="ab1 \t_|", min_size=0, max_size=10)))


@settings(max_examples=150, deadline=None)
@given(lines=st.lists(line, min_size=0, max_size=25), eol=st.sampled_from(["\n", "\r\n"]), tail=st.booleans(),
       threads=st.integers(1, 5))
def test_native_fasta_property(tmp_path_factory, lines, eol, tail, threads):
    data = eol.join(lines).encode("latin-1") + (eol.encode() if tail else b"")
    path = _write(tmp_path_factory.mktemp("fa"), "p.fa", data)
    recs = list(fasta.parse_sequences(path))
    buf = np.frombuffer(data, dtype=np.uint8)
    ix = _lib.fasta_index(buf)
    for pieces in (2, 3, 7):                             # the index cut into pieces at line starts and stitched
        again = _lib.fasta_index(buf, threads=pieces)
        assert all(np.array_equal(x, y) for x, y in zip(ix, again))
    assert ix[0].size == len(recs)
    assert ix[4].tolist() == [len(r.seq) for r in recs]
    assert [data[o:o + n].decode("latin-1") for o, n in zip(ix[0].tolist(), ix[1].tolist())] == [r.description for r in recs]
    if recs:
        codes, offsets = _lib.fasta_encode(buf, ix[2], ix[3], ix[4], 0, len(recs), pack._RNA_LUT, pack.SEP, threads)
        ref = pack.pack([pack.encode_rna(fasta.preprocess_seq(r.seq, True)) for r in recs])
        assert np.array_equal(codes, ref.codes) and np.array_equal(offsets, ref.offsets)


def test_lazy_fasta_over_several_files_and_compressed_input(tmp_path):
    import gzip
    a = _write(tmp_path, "a.fa", b">a1\nACGU\n>a2\nGG\n")
    b = _write(tmp_path, "b.fa", b">b1 x\nUUUU\nAA\n")
    with gzip.open(os.path.join(str(tmp_path), "c.fa.gz"), "wb") as f:
        f.write(b">c1\nACGT\n")
    lazy = fasta.LazyFasta([a, b])
    assert list(lazy.ids) == ["a1", "a2", "b1"] and list(lazy.lengths) == [4, 2, 6]
    codes, offsets, lengths = lazy[1:3].pack_rna()
    assert codes.tolist() == [2, 2, 7, 3, 3, 3, 3, 0, 0, 7] and offsets.tolist() == [0, 3] and lengths.tolist() == [2, 6]
    mixed = fasta.LazyFasta([a, os.path.join(str(tmp_path), "c.fa.gz")])
    assert list(mixed.ids) == ["a1", "a2", "c1"]
    assert mixed[0:3].pack_rna() is None                 # a compressed file has no mapped bytes: the Record path is taken
    assert mixed[0:2].pack_rna() is not None
    assert [r.seq for r in mixed] == ["ACGU", "GG", "ACGT"]


def test_cr_only_line_ends_take_the_record_path(tmp_path):
    """old-Mac line ends: the reference's text-mode reader breaks lines at a lone \\r, the native index does not -- such a
    file is left to the parsed-record path (CRLF files stay native)"""
    p = tmp_path / "mac.fa"
    p.write_bytes(b">b first\rACGU\rACGU\r>c\rGG\r")
    lz = fasta.LazyFasta(str(p))
    want = [(r.id, r.description, r.seq) for r in fasta.parse_sequences(str(p))]
    assert want == [("b", "b first", "ACGUACGU"), ("c", "c", "GG")]
    assert [(r.id, r.description, r.seq) for r in lz[0:len(lz)]] == want
    assert list(lz.lengths) == [8, 2]
    q = tmp_path / "dos.fa"
    q.write_bytes(b">b first\r\nACGU\r\nACGU\r\n")
    as_bytes = lambda path: np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    assert not _lib.fasta_lone_cr(as_bytes(q)) and _lib.fasta_lone_cr(as_bytes(p))
    assert [(r.id, r.seq) for r in fasta.LazyFasta(str(q))[0:1]] == [("b", "ACGUACGU")]
    # the WHOLE file is looked at (natively, in parallel pieces): one lone \r after 40 MB of \n / \r\n lines is found,
    # also as the very last byte and right at a piece boundary's \r\n
    big = tmp_path / "late.fa"
    body = b">r\nACGUACGUACGUACGUACGUACGUACGUACGUACGUACGU\r\nACGU\n" * 700000
    big.write_bytes(body + b">x\nAC\rGU\n")
    assert _lib.fasta_lone_cr(as_bytes(big)) and fasta.LazyFasta(str(big))._index[0] is None
    assert fasta.LazyFasta(str(big))[700000].seq == "ACGU"
    assert not _lib.fasta_lone_cr(np.frombuffer(body, dtype=np.uint8))
    assert _lib.fasta_lone_cr(np.frombuffer(body + b"\r", dtype=np.uint8)) and not _lib.fasta_lone_cr(np.zeros(0, dtype=np.uint8))


def test_fasta_index_argument_errors():
    L = _lib.load()
    assert L.pfmscan_fasta_index(None, 5, 0, None, None, None, None, None, None, 0) == _lib.E_BADARG
    buf = np.frombuffer(b">a\nAC\n>b\nGU\n", dtype=np.uint8)
    ix = _lib.fasta_index(buf)
    wrong = ix[4].copy()
    wrong[0] = 1                                         # an index that does not describe the bytes
    with pytest.raises(ValueError):
        _lib.fasta_encode(buf, ix[2], ix[3], wrong, 0, 2, pack._RNA_LUT)
    with pytest.raises(ValueError):
        _lib.fasta_encode(buf, ix[2], ix[3], ix[4], 0, 2, np.zeros(10, dtype=np.uint8))


def _float_strings(a):
    return "\n".join(a.astype(str).tolist()).replace("nan", "") + "\n"


@pytest.mark.parametrize("dtype,bits", [(np.float32, np.uint32), (np.float64, np.uint64)])
def test_float_fields_are_numpys_shortest_repr(dtype, bits):
    rng = np.random.default_rng(5)
    raw = rng.integers(0, np.iinfo(bits).max, size=200000, dtype=bits).view(dtype)            # every exponent, both signs
    with np.errstate(over="ignore"):
        special = np.array([0.0, -0.0, 1.0, -1.0, 6.0, 14.259, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9.999999e15, 123456.789, 0.001,
                            np.inf, -np.inf, np.nan, 5e-324, 1.7976931348623157e308, 1e22, 1e23, 0.1, 0.3, 2.5e-7], dtype=dtype)
    scores = np.round(rng.normal(0, 10, size=100000).astype(dtype), 3)                        # what LogOdds columns hold
    a = np.concatenate([special, scores, raw])
    got = b"".join(_lib.tsv_format([(_lib.TSV_F32 if dtype == np.float32 else _lib.TSV_F64, a, None, None, 0)], a.size)).decode()
    assert got == _float_strings(a)
    if dtype == np.float64:                                                                   # = repr(float), what to_csv writes
        fin = a[np.isfinite(a)][:5000]
        assert b"".join(_lib.tsv_format([(_lib.TSV_F64, fin, None, None, 0)], fin.size)).decode() == "".join(repr(float(x)) + "\n" for x in fin)


def test_writer_bytes_equal_to_csv(tmp_path):
    rng = np.random.default_rng(9)
    n = 5000
    ids = ["rec%d" % i for i in range(40)] + ['we"ird', "tab\there", "line\nbreak", "café", ""]
    rec = np.sort(rng.integers(0, len(ids), size=n))
    codes = rng.integers(0, 4, size=4000).astype(np.uint8)
    pos = rng.integers(0, 4000 - 12, size=n)
    f32 = np.round(rng.normal(0, 8, size=n).astype(np.float32), 3)
    f64 = rng.normal(0, 8, size=n)
    f64[::97] = np.nan
    f64[5] = np.inf
    cols = {"Sequence_ID": table.Indexed(ids, rec), "Description": table.Indexed([s + " desc" for s in ids], rec),
            "Motif_ID": "motif\tX", "Start": pos + 1, "End": pos + 12, "Sequence": table.Windows(codes, pos, 12, "ACGU"),
            "LogOdds": f32, "Struct": f64, "Sum": f32.astype(np.float64) + f64, "Dot": "."}
    df = table.to_frame(cols)
    assert df["Sequence"].tolist()[:3] == ["".join("ACGU"[c] for c in codes[p:p + 12]) for p in pos[:3]]
    want = df.copy()
    want["Match_ID"] = np.arange(1, n + 1)
    expected = want.to_csv(sep="\t", index=False)
    out = io.StringIO()
    w = table.TsvWriter(out, list(cols))
    w.write_chunk(cols)
    assert out.getvalue() == expected
    out = io.StringIO()                                  # the same table as a DataFrame, in uneven chunks
    table.write_frame(out, df, chunk=777)
    assert out.getvalue() == expected
    out = io.StringIO()                                  # more rows than one native call takes: cut inside write_chunk
    w = table.TsvWriter(out, list(cols))
    w.MAX_ROWS = 999
    w.write_chunk(cols)
    assert out.getvalue() == expected
    path = os.path.join(str(tmp_path), "o.tsv")          # a real text file: rows go to its byte buffer
    with open(path, "w", encoding="utf-8", newline="") as f:
        w = table.TsvWriter(f, list(cols))
        w.ASYNC_ROWS = 1000                                # the chunks of 1234 go out on the writer thread, the last one does not
        for lo in range(0, n, 1234):
            part = {k: (v if isinstance(v, str) else
                        table.Indexed(v.values, v.index[lo:lo + 1234]) if isinstance(v, table.Indexed) else
                        table.Windows(v.codes, v.pos[lo:lo + 1234], v.m, v.letters) if isinstance(v, table.Windows) else
                        v[lo:lo + 1234]) for k, v in cols.items()}
            w.write_chunk(part)
        w.close()
    with open(path, "r", encoding="utf-8", newline="") as f:
        assert f.read() == expected


def test_tsv_format_capacity_and_bad_descriptors():
    L = _lib.load()
    a = np.arange(1000, dtype=np.int64)
    text = _lib.tsv_format([(_lib.TSV_I64, a, None, None, 0)], a.size, 1, estimate=16)        # grows after E_CAPACITY
    assert b"".join(text).decode() == "".join("%d\t%d\n" % (i, i + 1) for i in range(1000))
    with pytest.raises(ValueError):
        _lib.tsv_format([(9, a, None, None, 0)], a.size)
    with pytest.raises(ValueError):
        _lib.tsv_format([(_lib.TSV_INDEXED, a, None, None, 0)], a.size)
    assert L.pfmscan_tsv_format(None, 1, 1, -1, None, 0, None, None, None, 0) == _lib.E_BADARG


def test_one_long_header_does_not_size_the_row_buffer_for_every_row():
    """the row buffer is sized from the values the rows really hold: one 100 kB FASTA header among 200 k rows must not ask
    for rows x 200 kB"""
    n = 200000
    blob = b"short" + b"x" * 100000
    spans = np.array([[0, 5], [5, 100000]], dtype=np.int64)
    index = np.zeros(n, dtype=np.int64)
    index[1234] = 1
    cols = [(_lib.TSV_SPAN, index, spans, blob, 0), (_lib.TSV_I64, np.arange(n, dtype=np.int64), None, None, 0)]
    box = [None]
    pieces = _lib.tsv_format(cols, n, first_match_id=1, estimate=1 << 16, scratch=box)
    text = b"".join(bytes(p) for p in pieces)
    assert box[0].size < 64 << 20                            # (it was n x (2 x 100000 + ...) = 40 GB before)
    lines = text.split(b"\n")
    assert len(lines) == n + 1 and lines[0] == b"short\t0\t1" and lines[1234] == b"x" * 100000 + b"\t1234\t1235"


def test_background_counts_natively_equal_the_record_loop(tmp_path):
    path = _write(tmp_path, "bg.fa", NASTY)
    want = {l: 0 for l in fasta.RNA}
    for rec in fasta.parse_sequences(path):
        s = fasta.preprocess_seq(rec.seq, True)
        for l in fasta.RNA:
            want[l] += s.count(l)
    got = fasta._count_rna_natively(path, positions=64)             # several batches
    assert {l: int(got[pack.RNA_LETTERS.index(l)]) for l in fasta.RNA} == want
    total = sum(want.values()) + 4
    bg = fasta.compute_background(path, fasta.RNA, verbose=False)
    assert bg == {l: (want[l] + 1.0) / total for l in fasta.RNA}


def test_ids_and_headers_as_spans_of_the_mapped_file(tmp_path):
    data = (b">r1 first\trecord \"q\"\nACGU\n>  \x1c r2\x1dmore  words \nGG\n>\nA\n>r4\r\nAC\n>we\"ird\nUU\n")
    path = _write(tmp_path, "s.fa", data)
    recs = list(fasta.parse_sequences(path))
    lazy = fasta.LazyFasta(path)
    buf, id_spans, hdr_spans = lazy[0:len(lazy)].span_tables()
    assert [buf[o:o + n].tobytes().decode() for o, n in id_spans.tolist()] == [r.id for r in recs]
    assert [buf[o:o + n].tobytes().decode() for o, n in hdr_spans.tolist()] == [r.description for r in recs]
    sub = lazy[1:4].span_tables()
    assert [buf[o:o + n].tobytes().decode() for o, n in sub[1].tolist()] == [r.id for r in recs[1:4]]
    # the writer quotes span fields like to_csv does
    idx = np.array([0, 0, 1, 2, 3, 4, 4])
    cols = {"Sequence_ID": table.Spans(buf, id_spans, idx), "Description": table.Spans(buf, hdr_spans, idx), "Start": np.arange(7)}
    df = table.to_frame(cols)
    assert df["Sequence_ID"].tolist() == [recs[i].id for i in idx] and df["Description"].tolist() == [recs[i].description for i in idx]
    out = io.StringIO()
    table.TsvWriter(out, list(cols), match_id=False).write_chunk(cols)
    assert out.getvalue() == df.to_csv(sep="\t", index=False)
    # a header with a non-ASCII byte: no spans, the strings are decoded in Python as before
    path2 = _write(tmp_path, "n.fa", NASTY)
    assert fasta.LazyFasta(path2)[0:3].span_tables() is None
    two = fasta.LazyFasta([path, path])
    assert two[3:7].span_tables() is None and two[5:8].span_tables() is not None        # a slice across two files has no one buffer


def test_bulk_ids_equal_the_lazy_ones(tmp_path):
    data = b">r1 first\nAC\n>  r2\tx \nGG\n>\nA\n>r4\r\nAC\n"
    path = _write(tmp_path, "b.fa", data)
    recs = list(fasta.parse_sequences(path))
    lazy = fasta.LazyFasta(path)
    assert lazy.ids.tolist() == [r.id for r in recs] and lazy.headers.tolist() == [r.description for r in recs]
    assert lazy.ids[1:3] == [r.id for r in recs[1:3]] and list(lazy[1:4].ids) == [r.id for r in recs[1:4]]
    assert [lazy.ids[i] for i in range(4)] == [r.id for r in recs]
    nasty = fasta.LazyFasta(_write(tmp_path, "n.fa", NASTY))            # a non-ASCII header: decoded one by one as before
    assert nasty.ids.tolist() == [r.id for r in fasta.parse_sequences(os.path.join(str(tmp_path), "n.fa"))]
    assert _lib.gather_spans(np.frombuffer(b"abcdef", dtype=np.uint8), np.array([[1, 2], [4, 0], [3, 3]])) == b"bc\n\ndef\n"


def test_index_of_very_short_records_takes_the_exact_count_retry():
    data = b"".join(b">r%d\nAC\n" % i for i in range(5000))           # 8-9 bytes per record: more records than size // 64
    buf = np.frombuffer(data, dtype=np.uint8)
    ix = _lib.fasta_index(buf)
    assert ix[0].size == 5000 and ix[4].tolist() == [2] * 5000
    assert data[ix[0][4999]:ix[0][4999] + ix[1][4999]] == b"r4999"


def _pandas_profile(path):
    df = pd.read_table(path)
    if "PO" in df.columns:
        del df["PO"]
    else:
        df = df.iloc[:, 1:]
    return [str(c) for c in df.columns], df.to_numpy(dtype=np.float64).reshape(-1, len(df.columns))


def _write_profile(path, values, eol="\n", fmt=repr, header="PO\t" + "\t".join("BEHLMRT")):
    with open(path, "w", newline="") as f:
        f.write(header + eol)
        for i, row in enumerate(values):
            f.write(str(i) + "".join("\t" + fmt(float(v)) for v in row) + eol)


def test_native_profile_parser_is_pandas_bit_for_bit(tmp_path):
    """the reference reads its profiles with pd.read_table (rnascan.py:296-297), whose default float converter is not
    correctly rounded; pfmscan_profile_parse restates that converter: every field must come out with pandas' bits"""
    from conftest import DATA_DIR
    rng = np.random.default_rng(17)
    cases = {
        "dirichlet": rng.dirichlet(np.full(7, 0.3), size=8000),
        "uniform": rng.random((8000, 7)),
        "tiny": 10.0 ** rng.uniform(-320, -3, size=(5000, 7)),
        "huge": 10.0 ** rng.uniform(3, 300, size=(5000, 7)) * rng.choice([-1.0, 1.0], size=(5000, 7)),
        "short": np.round(rng.random((2000, 7)), 3),
        "ints": rng.integers(0, 3, size=(2000, 7)).astype(float),
    }
    differs_from_float = 0
    for name, x in cases.items():
        for eol in ("\n", "\r\n"):
            path = str(tmp_path / (name + ".txt"))
            _write_profile(path, x, eol)
            letters, got = fasta.read_profile(path)
            want_letters, want = _pandas_profile(path)
            assert letters == want_letters == list("BEHLMRT")
            assert got.shape == want.shape and np.array_equal(got.view(np.uint64), want.view(np.uint64)), name
            differs_from_float += int((want != x).sum())
    assert differs_from_float > 30000             # the converter IS different from float(): that is what is being matched
    path = os.path.join(DATA_DIR, "HIST2H3C_3p_end_structure.txt")      # the reference's own example profile
    letters, got = fasta.read_profile(path)
    want_letters, want = _pandas_profile(path)
    assert letters == want_letters and np.array_equal(got.view(np.uint64), want.view(np.uint64))
    data = open(path, "rb").read()
    assert _lib.profile_parse(data, len(letters)) is not None             # and it did go through the native parser


def test_unusual_profile_files_are_left_to_pandas(tmp_path):
    x = np.random.default_rng(3).random((50, 7))
    odd = {
        "nan token": lambda p: _write_profile(p, x, fmt=lambda v: "nan" if v < 0.05 else repr(v)),
        "blank line": lambda p: (_write_profile(p, x), open(p, "a").write("\n\n")),
        "no PO header": lambda p: _write_profile(p, x, header="idx\t" + "\t".join("BEHLMRT")),
        # 'PO' is deleted BY NAME (rnascan.py:297): when it is not the first column, the first column is a letter's
        "PO in the second column": lambda p: _write_profile(p, x, header="E\tPO\t" + "\t".join("HTBLRM")),
        "exponent forms": lambda p: _write_profile(p, x, fmt=lambda v: "%.6E" % v),
        "plus signs": lambda p: _write_profile(p, x, fmt=lambda v: "+" + repr(v)),
    }
    for name, write in odd.items():
        path = str(tmp_path / "odd.txt")
        write(path)
        letters, got = fasta.read_profile(path)
        want_letters, want = _pandas_profile(path)
        assert letters == want_letters, name
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), name
    ragged = str(tmp_path / "ragged.txt")
    _write_profile(ragged, x)
    with open(ragged, "a") as f:
        f.write("50\t0.1\t0.2\n")
    assert _lib.profile_parse(open(ragged, "rb").read(), 7) is None
    # a batch of files on the thread pool: same results, same order
    paths = []
    for k in range(12):
        p = str(tmp_path / ("structure.r%d.txt" % k))
        _write_profile(p, np.random.default_rng(k).dirichlet(np.full(7, 0.3), size=10 + 7 * k))
        paths.append(p)
    many = fasta.read_profiles(paths, threads=4)
    for p, (letters, prof) in zip(paths, many):
        want_letters, want = _pandas_profile(p)
        assert letters == want_letters and np.array_equal(prof.view(np.uint64), want.view(np.uint64))


def test_ingest_under_sanitizers(tmp_path):
    """the host-only native code (pfmscan_ingest.hip has no device code) compiled with g++ -fsanitize=address,undefined
    and driven with random bytes in exact-size heap buffers (tests/c/fuzz_ingest.cpp): no overread, no overflow, no UB"""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    src = open(os.path.join(root, "rnascan_amd", "csrc", "pfmscan_ingest.hip")).read()
    assert '#include "pfmscan_ctx.hpp"' in src
    src = src.replace('#include "pfmscan_ctx.hpp"',
                      '#include <cstdint>\n#include <string>\n#include "pfmscan.h"\n'
                      'namespace pfmscan { int fail(pfmscan_ctx *ctx, int code, const std::string &msg); }')
    (tmp_path / "ingest.cpp").write_text(src)
    (tmp_path / "stub.cpp").write_text('#include <string>\nstruct pfmscan_ctx;\n'
                                       'namespace pfmscan { int fail(pfmscan_ctx *, int code, const std::string &) { return code; } }\n')
    exe = str(tmp_path / "fuzz")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(root, "include"), "-pthread", os.path.join(here, "c", "fuzz_ingest.cpp"),
           str(tmp_path / "ingest.cpp"), str(tmp_path / "stub.cpp"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert built.returncode == 0, built.stderr[-2000:]
    run = subprocess.run([exe, "4000"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and run.stdout.strip().startswith("ok"), (run.stdout + run.stderr)[-3000:]


def test_staged_upload_source_must_still_be_the_mapped_file(tmp_path):
    """pfmscan_upload_source_file_checked: the path is pread in the mapping's place only while it names the file that was
    mapped (st_dev / st_ino / st_size recorded at mapping time) -- a store re-packed by rename is refused.  The check sits
    in front of any device work, so it runs without a GPU on a NULL context: BADARG either way, the message tells which."""
    import ctypes
    L = _lib.load()
    p = tmp_path / "rows.bin"
    p.write_bytes(b"x" * 4096)
    st = os.stat(p)
    buf = np.zeros(4096, dtype=np.uint8)
    rc = L.pfmscan_upload_source_file_checked(None, buf.ctypes.data_as(ctypes.c_void_p), 4096, os.fsencode(str(p)), 0, st.st_dev, st.st_ino, st.st_size)
    assert rc == _lib.E_BADARG                     # no context: refused before the file is looked at


def test_count_bytes_and_background_of_a_structure_fasta(tmp_path):
    """compute_background (rnascan.py:440-465) over a structure FASTA: counted from the packed codes by the native byte
    histogram -- upper-case letters only, as Seq.count does on the untouched record (rnascan.py:186-197) -- and equal to
    the parsed-record path"""
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, size=1_000_003).astype(np.uint8)
    assert np.array_equal(_lib.count_bytes(x), np.bincount(x, minlength=256)) and _lib.count_bytes(x[:0]).sum() == 0
    p = tmp_path / "t.fa"
    with open(p, "w") as f:
        for i in range(50):
            f.write(">r%d\n%s\n" % (i, "".join(rng.choice(list("EHTBLRMehtblrmX"), size=int(rng.integers(0, 300))))))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        native = fasta.compute_background(str(p), fasta.STRUCT, verbose=False)
        gz = tmp_path / "t.fa.gz"
        import gzip
        gz.write_bytes(gzip.compress(p.read_bytes()))
        parsed = fasta.compute_background(str(gz), fasta.STRUCT, verbose=False)        # compressed: the parsed-record path
    assert native == parsed and abs(sum(native.values()) - 1.0) < 1e-12
    assert fasta.open_lazy(str(p)) is fasta.open_lazy(str(p))                          # indexed once
    p.write_text(">only\nEEE\n")
    assert len(fasta.open_lazy(str(p))) == 1                                           # a changed file is indexed again
