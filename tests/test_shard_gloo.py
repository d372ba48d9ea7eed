"""N>1 path on CPU: two gloo ranks each scan their contiguous record range (scores from
the TEST-ONLY OracleEngine), rank 0 gathers; the result must equal the unsharded table."""
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest

from conftest import DATA_DIR, REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_inputs():
    from rnascan_amd import fasta, pssm
    rng = np.random.default_rng(42)
    recs = []
    for i in range(37):
        L = int(rng.integers(0, 400))
        recs.append(fasta.Record("r%d" % i, "r%d some description" % i, "".join(rng.choice(list("ACGTN"), size=L, p=[.24, .24, .24, .24, .04]))))
    P = {"SLBP": pssm.pfm2pssm(os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt"), 0.01, fasta.RNA, None)}
    return recs, P


def _worker(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from engines import OracleEngine
    from rnascan_amd import fasta, scanner, shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    recs, P = _make_inputs()
    eng = OracleEngine()
    table = shard.scan_sharded(recs, [len(r.seq) for r in recs],
                               lambda part: scanner.scan_records(eng, part, P, fasta.RNA, -2.0),
                               rank=rank, world=world, dist=dist)
    if rank == 0:
        scanner._add_match_id(table)
        table.to_csv(os.path.join(outdir, "sharded.tsv"), sep="\t", index=False)
    else:
        assert table is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_scan_equals_single_rank(tmp_path):
    import torch.multiprocessing as mp
    from engines import OracleEngine
    from rnascan_amd import fasta, scanner
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    recs, P = _make_inputs()
    single = scanner.scan_records(OracleEngine(), recs, P, fasta.RNA, -2.0)
    scanner._add_match_id(single)
    want = single.to_csv(sep="\t", index=False)
    got = open(tmp_path / "sharded.tsv").read()
    assert len(single) > 20
    assert got == want


def _cli_worker(rank, world, port, outdir, argv):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RNASCAN_DIST_BACKEND": "gloo"})
    from engines import OracleEngine
    from rnascan_amd import cli
    with open(os.path.join(outdir, "out.%d.tsv" % rank), "w") as out:
        cli.main(argv, engine=OracleEngine(), out=out)
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_cli_under_two_gloo_ranks(tmp_path):
    """the drop-in CLI launched torchrun-style: rank 0 prints the whole table, rank 1 nothing"""
    import io
    import torch.multiprocessing as mp
    from engines import OracleEngine
    from rnascan_amd import cli
    rng = np.random.default_rng(7)
    fa = tmp_path / "many.fa"
    with open(fa, "w") as f:
        for i in range(23):
            f.write(">rec%d desc %d\n%s\n" % (i, i, "".join(rng.choice(list("ACGT"), size=int(rng.integers(30, 500))))))
    argv = ["-p", os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt"), "-C", "0.01", "-m", "-1", str(fa)]
    mp.spawn(_cli_worker, args=(2, _free_port(), str(tmp_path), argv), nprocs=2, join=True)
    single = io.StringIO()
    cli.main(argv, engine=OracleEngine(), out=single)
    assert open(tmp_path / "out.0.tsv").read() == single.getvalue()
    assert open(tmp_path / "out.1.tsv").read() == ""
    assert single.getvalue().count("\n") > 10


def test_batches_cover_the_range_in_order():
    from rnascan_amd import shard
    lengths = [5, 0, 9, 100, 3, 3, 3, 50]
    for cap in (1, 4, 10, 12, 60, 10 ** 6):
        parts = shard.batches(lengths, 1, 8, cap)
        assert parts[0][0] == 1 and parts[-1][1] == 8
        assert all(a < b for a, b in parts) and all(p[1] == q[0] for p, q in zip(parts, parts[1:]))
        for a, b in parts:                                   # over the cap only when a single record is
            assert sum(l + 1 for l in lengths[a:b]) <= cap or b - a == 1
    assert shard.batches(lengths, 3, 3, 10) == [(3, 3)]


def test_batched_scan_equals_one_launch(monkeypatch):
    """RNASCAN_BATCH_POSITIONS bounds what one launch holds; the table does not change"""
    from engines import OracleEngine
    from rnascan_amd import fasta, scanner, shard
    recs, P = _make_inputs()
    eng = OracleEngine()
    fn = lambda part: scanner.scan_records(eng, part, P, fasta.RNA, -2.0)   # noqa: E731
    whole = shard.scan_sharded(recs, [len(r.seq) for r in recs], fn, rank=0, world=1)
    calls = []
    fn2 = lambda part: (calls.append(len(part)), fn(part))[1]               # noqa: E731
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "700")
    cut = shard.scan_sharded(recs, [len(r.seq) for r in recs], fn2, rank=0, world=1)
    assert len(calls) > 5 and sum(calls) == len(recs)
    pd.testing.assert_frame_equal(whole.reset_index(drop=True), cut.reset_index(drop=True))


def test_sink_receives_the_batches_in_order(monkeypatch):
    """one rank + sink: nothing is returned, the batches' tables arrive one by one, in record order"""
    from engines import OracleEngine
    from rnascan_amd import fasta, scanner, shard
    recs, P = _make_inputs()
    eng = OracleEngine()
    fn = lambda part: scanner.scan_records(eng, part, P, fasta.RNA, -2.0)   # noqa: E731
    whole = shard.scan_sharded(recs, [len(r.seq) for r in recs], fn, rank=0, world=1)
    got = []
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "1500")
    ret = shard.scan_sharded(recs, [len(r.seq) for r in recs], fn, rank=0, world=1, sink=got.append)
    assert ret is None and len(got) > 3
    cat = pd.concat([g for g in got if len(g)], ignore_index=True)
    pd.testing.assert_frame_equal(whole.reset_index(drop=True), cat)


def _write_rnass_inputs(tmp_path, n=19, seed=3):
    """a FASTA + a directory of averaged-structure profiles (pfmutil.py:61-87 format): most records pair one to one,
    one has no profile, one profile has no record, one profile is SHORTER than its record (the batch holding it takes
    the two-table join)"""
    rng = np.random.default_rng(seed)
    fa = tmp_path / "seqs.fa"
    d = tmp_path / "avg"
    d.mkdir()

    def write_profile(sid, L):
        with open(d / ("structure.%s.txt" % sid), "w") as f:
            f.write("PO\t" + "\t".join("BEHLMRT") + "\n")
            p = rng.dirichlet(np.full(7, 0.3), size=L)
            p[p < 0.02] = 0.0
            for i in range(L):
                f.write(str(i) + "\t" + "\t".join(repr(float(x)) for x in p[i]) + "\n")

    with open(fa, "w") as f:
        for i in range(n):
            L = int(rng.integers(20, 260))
            f.write(">rec%d the %dth record\n%s\n" % (i, i, "".join(rng.choice(list("ACGT"), size=L))))
            if i == 4:
                continue                                   # no profile for this record
            write_profile("rec%d" % i, L - 7 if i == 9 else L)
    write_profile("orphan", 50)
    return str(fa), str(d)


@pytest.mark.timeout(600)
def test_cli_rnass_directory_and_store_under_two_gloo_ranks(tmp_path, monkeypatch):
    """sequence FASTA + averaged-structure directory (config 3's command line), and the same through a packed profile
    store: two ranks (each reading only its share of records and profiles) print what one rank prints"""
    import io
    import torch.multiprocessing as mp
    from engines import OracleEngine
    from rnascan_amd import cli, store
    fa, d = _write_rnass_inputs(tmp_path)
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "900")    # several batches per rank
    base = ["-p", os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt"),
            "-q", os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", "-m", "-30"]
    st = tmp_path / "packed"
    store.build_store(d, str(st))
    for tag, struct_arg in (("dir", d), ("store", str(st))):
        argv = base + [fa, struct_arg]
        single = io.StringIO()
        cli.main(argv, engine=OracleEngine(), out=single)
        assert single.getvalue().count("\n") > 30
        out = tmp_path / tag
        out.mkdir()
        mp.spawn(_cli_worker, args=(2, _free_port(), str(out), argv), nprocs=2, join=True)
        assert open(out / "out.0.tsv").read() == single.getvalue()
        assert open(out / "out.1.tsv").read() == ""
    # the two inputs describe the same profiles: same table up to the float64 text round trip of the store
    # (directory and store rows come in FASTA order either way)


def test_rnass_batches_equal_the_reference_join(tmp_path):
    """the per-batch fused / fallback path == combine(scan_main(fasta), scan_main(dir)) over everything"""
    import io
    from engines import OracleEngine
    from rnascan_amd import cli, fasta, pssm, scanner
    fa, d = _write_rnass_inputs(tmp_path, n=12, seed=8)
    base = ["-p", os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt"),
            "-q", os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", "-m", "-30", fa, d]
    got = io.StringIO()
    cli.main(base, engine=OracleEngine(), out=got)
    eng = OracleEngine()
    ps = pssm.load_pssms(base[1], 0.01, fasta.RNA, None)
    pt = pssm.load_pssms(base[3], 0.01, fasta.STRUCT, None)
    seq = scanner.scan_records(eng, list(fasta.parse_sequences(fa)), ps, fasta.RNA, -30.0)
    named = scanner.load_profile_dir(d)
    order = {r.id: i for i, r in enumerate(fasta.parse_sequences(fa))}
    named.sort(key=lambda t: order.get(t[0], 10 ** 9))
    stt = scanner.scan_profiles(eng, named, pt, -30.0, "aligned", np.float64)
    want = scanner.combine(seq, stt)
    scanner._add_match_id(want)
    assert got.getvalue() == want.to_csv(sep="\t", index=False)
    assert len(want) > 30


def _cli_worker_no_gather(rank, world, port, outdir, argv):
    """like _cli_worker, but any attempt to gather tables (pickled DataFrames through gather_object) fails the rank"""
    sys.path.insert(0, REPO)
    from rnascan_amd import shard

    def boom(*a, **k):
        raise AssertionError("a rank tried to gather whole tables")
    shard.gather_frames = boom
    _cli_worker(rank, world, port, outdir, argv)


@pytest.mark.timeout(600)
def test_two_ranks_stream_their_rows_and_rank0_never_holds_a_table(tmp_path, monkeypatch):
    """-m ' -inf' (every window a row) under two ranks: each rank streams its batches through the native row formatter;
    rank 1's rows reach rank 0 as BYTES (spool file), rank 0 numbers Match_ID while it copies -- no gather of tables, and
    the bytes equal the single-rank output.  Headers with tabs and quotes go through too (csv quoting survives the relay)."""
    import io
    import torch.multiprocessing as mp
    from engines import OracleEngine
    from rnascan_amd import cli
    rng = np.random.default_rng(11)
    fa = tmp_path / "all.fa"
    with open(fa, "w") as f:
        for i in range(31):
            extra = ' with a "quote" and a\ttab' if i % 5 == 0 else ""
            f.write(">rec%d desc %d%s\n%s\n" % (i, i, extra, "".join(rng.choice(list("ACGT"), size=int(rng.integers(20, 300))))))
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "100")     # several batches per rank (the RNA path takes 8 x this)
    monkeypatch.setenv("RNASCAN_SPOOL_DIR", str(tmp_path))
    argv = ["-p", os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt"), "-C", "0.01", "-m", " -inf", str(fa)]
    single = io.StringIO()
    cli.main(argv, engine=OracleEngine(), out=single)
    assert single.getvalue().count("\n") > 2000
    for send in ("0", "1"):                                 # rank 0 reads the spool file / the rows travel over the process group
        monkeypatch.setenv("RNASCAN_SPOOL_SEND", send)
        mp.spawn(_cli_worker_no_gather, args=(3 if send == "1" else 2, _free_port(), str(tmp_path), argv), nprocs=3 if send == "1" else 2,
                 join=True)
        assert open(tmp_path / "out.0.tsv").read() == single.getvalue()
        assert open(tmp_path / "out.1.tsv").read() == ""
        assert not [p for p in os.listdir(tmp_path) if p.startswith("rnascan_rows_")]        # the spools are gone


def _write_fasta_pair(tmp_path, n=23, seed=5):
    """a sequence FASTA and a structure FASTA with the same record ids in the same order (mixed-case structure strings,
    some foreign letters, headers that need csv quoting)"""
    rng = np.random.default_rng(seed)
    fa, fb = tmp_path / "seqs.fa", tmp_path / "structs.fa"
    with open(fa, "w") as f, open(fb, "w") as g:
        for i in range(n):
            L = int(rng.integers(0, 400))
            s = "".join(rng.choice(list("ACGTUacgtN"), size=L))
            t = "".join(rng.choice(list("EHTBLRMehtblrmx"), size=L))
            f.write(">rec%d seq %d\n" % (i, i) + "\n".join(s[k:k + 60] for k in range(0, max(L, 1), 60)) + "\n")
            g.write(">rec%d struct%s\n" % (i, ' "q"\ttab' if i % 4 == 0 else "") + "\n".join(t[k:k + 45] for k in range(0, max(L, 1), 45)) + "\n")
    return str(fa), str(fb)


@pytest.mark.timeout(600)
def test_two_fasta_rnass_streams_under_two_ranks_without_table_gathers(tmp_path, monkeypatch):
    """`rnascan -p .. -q .. seqs.fa structs.fa` (rnascan.py:119-123): one rank prints combine() of the two single tables
    (rnascan.py:416-434, made here through the reference-shaped two-table path); two ranks stream the fused scan's rows
    through their spools -- any gather of tables fails the run -- and give the same bytes."""
    import io
    import torch.multiprocessing as mp
    from engines import OracleEngine
    from rnascan_amd import cli
    fa, fb = _write_fasta_pair(tmp_path)
    pfm_s, pfm_t = tmp_path / "s.pfm", tmp_path / "t.pfm"
    rng = np.random.default_rng(2)
    for path, letters in ((pfm_s, "ACGU"), (pfm_t, "EHTBLRM")):
        with open(path, "w") as f:
            f.write("PO\t" + "\t".join(letters) + "\n")
            for j in range(5):
                f.write("%d\t" % j + "\t".join("%.4f" % x for x in rng.dirichlet(np.full(len(letters), 0.5))) + "\n")
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "150")
    monkeypatch.setenv("RNASCAN_SPOOL_DIR", str(tmp_path))
    for minscore in ("-1.5", " -inf"):
        argv = ["-p", str(pfm_s), "-q", str(pfm_t), "-u", "-C", "0.01", "-m", minscore, fa, fb]
        want = io.StringIO()
        with monkeypatch.context() as mp_ctx:
            mp_ctx.setattr(cli, "_same_records", lambda *a: None)           # two tables + combine(), as the reference does it
            cli.main(argv, engine=OracleEngine(), out=want)
        single = io.StringIO()
        cli.main(argv, engine=OracleEngine(), out=single)
        assert single.getvalue() == want.getvalue() and want.getvalue().count("\n") > 20
        mp.spawn(_cli_worker_no_gather, args=(2, _free_port(), str(tmp_path), argv), nprocs=2, join=True)
        assert open(tmp_path / "out.0.tsv").read() == want.getvalue()
        assert open(tmp_path / "out.1.tsv").read() == ""


def test_tsv_number_respects_quoted_line_breaks_across_blocks():
    from rnascan_amd import _lib
    text = b'a\t1\n"x ""q""\ny"\t2\nlast\t3\n'
    want = b'a\t1\t7\n"x ""q""\ny"\t2\t8\nlast\t3\t9\n'
    got, rows, state = _lib.tsv_number(text, 7)
    assert (got, rows, state) == (want, 3, 0)
    for cut in range(1, len(text)):                        # any split into two blocks gives the same bytes
        a, ra, st = _lib.tsv_number(text[:cut], 7)
        b, rb, st = _lib.tsv_number(text[cut:], 7 + ra, st)
        assert a + b == want and ra + rb == 3 and st == 0
    assert _lib.tsv_number(b"", 1) == (b"", 0, 0)


def test_relay_into_a_text_sink_survives_blocks_that_split_a_character(tmp_path, monkeypatch):
    """rank 0's side of relay_spools with a text sink without .buffer: the spool is read in blocks that end inside
    multi-byte characters; the table comes out intact and numbered"""
    import io
    from rnascan_amd import shard
    rows = ["récord %d ☃ désc\t%d" % (i, i) for i in range(40)]
    spool = tmp_path / "rank1.tsv"
    spool.write_bytes("".join(r + "\n" for r in rows).encode("utf-8"))

    class OneNode(object):                                  # the collectives of a 2-rank group as rank 0 sees them
        @staticmethod
        def gather_object(obj, bucket, dst=0):
            bucket[0], bucket[1] = obj, (str(spool), len(rows))

        @staticmethod
        def broadcast_object_list(box, src=0):
            pass

        @staticmethod
        def barrier():
            pass
    monkeypatch.setattr(shard, "SPOOL_BLOCK", 7)
    out = io.StringIO()
    nxt = shard.relay_spools(out, 5, None, 0, 0, 2, OneNode)
    assert nxt == 45
    assert out.getvalue() == "".join("%s\t%d\n" % (r, 5 + i) for i, r in enumerate(rows))
