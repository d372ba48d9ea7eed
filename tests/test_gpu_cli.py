"""End to end on the GPU: the drop-in CLI and the scanner layer with the real HipEngine,
against the reference's captured tables (config 1: example/ HIST2H3C + SLBP PFMs)."""
import io
import os
import shutil

import numpy as np
import pytest

from conftest import DATA_DIR, nasty_fasta

pytestmark = pytest.mark.gpu

SEQ_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt")
STRUCT_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_struct.txt")
HIST_FA = os.path.join(DATA_DIR, "HIST2H3C_3p_end.fa")
HIST_PROFILE = os.path.join(DATA_DIR, "HIST2H3C_3p_end_structure.txt")


@pytest.fixture(scope="module")
def engine():
    from rnascan_amd import scanner
    e = scanner.HipEngine(0)
    yield e
    e.close()


@pytest.fixture()
def workdir(tmp_path):
    d = tmp_path / "avg"
    d.mkdir()
    shutil.copyfile(HIST_PROFILE, d / "structure.hg19_dna.txt")
    os.symlink(SEQ_PFM, tmp_path / "SLBP_seq.txt")
    os.symlink(STRUCT_PFM, tmp_path / "SLBP_struct.txt")
    (tmp_path / "bg_struct.txt").write_text(repr({l: 1.0 / 7 for l in "EHTBLRM"}))
    return tmp_path


def test_cli_combined_matches_reference_table(engine, golden, workdir):
    from rnascan_amd import cli
    from test_scanner_cpu import assert_tsv_equal
    out = io.StringIO()
    cli.main(["-p", str(workdir / "SLBP_seq.txt"), "-q", str(workdir / "SLBP_struct.txt"), "-C", "0.01", "-m", "0",
              "-B", str(workdir / "bg_struct.txt"), HIST_FA, str(workdir / "avg")], engine=engine, out=out)
    assert_tsv_equal(out.getvalue(), golden["combine"]["tsv"])
    # float32 profile storage (the benchmark's layout): same table within 1e-6
    out32 = io.StringIO()
    cli.main(["-p", str(workdir / "SLBP_seq.txt"), "-q", str(workdir / "SLBP_struct.txt"), "-C", "0.01", "-m", "0",
              "-B", str(workdir / "bg_struct.txt"), "--profile-dtype", "float32", HIST_FA, str(workdir / "avg")],
             engine=engine, out=out32)
    assert_tsv_equal(out32.getvalue(), golden["combine"]["tsv"], tol=1e-6)


def test_config1_minus_inf_all_219_windows(engine, golden):
    """BASELINE.json configs[0]: -m ' -inf' on the example inputs, both pairings"""
    from rnascan_amd import fasta, pssm, scanner
    rec = list(fasta.parse_sequences(HIST_FA))[0]
    sp = {"SLBP": pssm.pfm2pssm(SEQ_PFM, 0.0, fasta.RNA, None)}
    df = scanner.scan_records(engine, [rec], sp, fasta.RNA, float("-inf"))
    g = [c for c in golden["pwm"] if c["name"] == "hist_slbp_pc0_uniform"][0]
    assert len(df) == 219 and df["Start"].tolist() == list(range(1, 220))
    assert np.array_equal(df["LogOdds"].to_numpy(), np.round(np.array(g["scores"], dtype=np.float32), 3))
    assert df["Sequence"].iloc[212] == "AAAGGCUCUUUUCAGAGC" and float(df["LogOdds"].iloc[212]) == float(np.float32(14.259))
    tp = {"SLBP": pssm.pfm2pssm(STRUCT_PFM, 0.0, fasta.STRUCT, None)}
    cases = {c["name"]: c for c in golden["scan_averaged_structure"]}
    for pairing, name in (("positional", "hist_slbp_pc0_positional"), ("aligned", "hist_slbp_pc0_aligned")):
        d = scanner.scan_averaged_structure(engine, HIST_PROFILE, tp, float("-inf"), pairing)
        want = cases[name]["rows"]
        assert d["Start"].tolist() == [r[0] for r in want]
        assert np.abs(d["LogOdds"].to_numpy() - np.array([r[2] for r in want])).max() <= 1e-9


def test_scanner_layer_gpu_equals_oracle_engine(engine):
    """same tables from the HIP engine and the test-only oracle engine on ragged random records"""
    import pandas as pd
    from engines import OracleEngine
    from rnascan_amd import fasta, pssm, scanner
    rng = np.random.default_rng(11)
    recs, named = [], []
    for i in range(40):
        L = int(rng.integers(0, 900))
        recs.append(fasta.Record("id%d" % i, "id%d d" % i, "".join(rng.choice(list("ACGTNacgu"), size=L))))
        p = rng.dirichlet(np.full(7, 0.3), size=L) if L else np.zeros((0, 7))
        p[p < 0.02] = 0
        if i % 5:
            named.append(("id%d" % i, list("BEHLMRT"), p))
    sp = {"a": pssm.pfm2pssm(SEQ_PFM, 0.0, fasta.RNA, None)}
    tp = {"b": pssm.pfm2pssm(STRUCT_PFM, 0.0, fasta.STRUCT, None)}
    for thr in (float("-inf"), -3.0, 2.0):
        a = scanner.scan_combined(engine, recs, named, sp, tp, thr, "aligned", np.float64)
        b = scanner.scan_combined(OracleEngine(), recs, named, sp, tp, thr, "aligned", np.float64)
        assert len(a) == len(b)
        pd.testing.assert_frame_equal(a.drop(columns=["LogOdds.Struct", "LogOdds.SeqStruct"]),
                                      b.drop(columns=["LogOdds.Struct", "LogOdds.SeqStruct"]))
        for col in ("LogOdds.Struct", "LogOdds.SeqStruct"):
            x, y = a[col].to_numpy(), b[col].to_numpy()
            big = np.abs(y) > 1e9
            assert np.abs(x[~big] - y[~big]).max(initial=0) <= 1e-6 and np.allclose(x[big], y[big], rtol=1e-12)
        s1 = scanner.scan_records(engine, recs, sp, fasta.RNA, thr)
        s2 = scanner.scan_records(OracleEngine(), recs, sp, fasta.RNA, thr)
        pd.testing.assert_frame_equal(s1, s2)


def test_multi_pfm_library_on_gpu_reuses_the_staged_stream(engine, tmp_path):
    """N1 on the device: the packed stream is uploaded once (ctx scratch generation does not
    move between motifs) and every motif's table equals the oracle engine's"""
    import pandas as pd
    from engines import OracleEngine
    from rnascan_amd import fasta, pssm, scanner
    from test_scanner_cpu import _write_multi_pfm
    rng = np.random.default_rng(3)
    motifs = [("M%03d" % k, list("ACGU"), rng.dirichlet(np.full(4, 0.5), size=int(rng.integers(6, 19)))) for k in range(8)]
    lib = str(tmp_path / "lib.pfm")
    _write_multi_pfm(lib, motifs)
    P = pssm.load_pssms(lib, 0.01, fasta.RNA, None)
    recs = [fasta.Record("r%d" % i, "r%d" % i, "".join(rng.choice(list("ACGTN"), size=int(rng.integers(0, 2000)))))
            for i in range(50)]
    g0 = engine.ctx.scratch_gen
    got = scanner.scan_records(engine, recs, P, fasta.RNA, 2.0)
    assert engine.ctx.scratch_gen == g0 + 1                     # one upload for 8 motifs
    want = scanner.scan_records(OracleEngine(), recs, P, fasta.RNA, 2.0)
    pd.testing.assert_frame_equal(got, want)
    assert got["Motif_ID"].nunique() == 8


def test_staged_api_matches_host_api(ctx, oracle):
    from rnascan_amd import pack
    from test_gpu_parity import rand_stream, rand_table, rand_struct_pssm
    from conftest import assert_f32_bits_equal, assert_struct_close
    rng = np.random.default_rng(77)
    s = rand_stream(rng, 30, 0, 800)
    ctx.stage(s.codes, s.profile)
    for m in (5, 12, 20):
        T, P = rand_table(rng, m), rand_struct_pssm(rng, m, inf_frac=0.1)
        motif = ctx.motif(T, P)
        sq, st = ctx.scan_staged(motif)
        assert_f32_bits_equal(sq, oracle.stream_seq(s.codes, T))
        assert_struct_close(st, oracle.stream_struct(s.profile, P))
        pos, hs, hst = ctx.hits_staged(motif, 0.0, -40.0)
        want = oracle.stream_hits(oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P), 0.0, -40.0)
        assert np.array_equal(pos, want)
        motif.close()
    with pytest.raises(ValueError):
        only_seq = ctx.motif(letter_table=rand_table(rng, 4))
        ctx.stage(None, s.profile)
        ctx.hits_staged(only_seq, 0.0, 0.0)


def _gpu_cli_worker(rank, world, port, outdir, argv):
    import sys
    from conftest import REPO
    sys.path.insert(0, REPO)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "RNASCAN_ONE_DEVICE": "1"})   # rehearsal on a one-GPU box: every rank on --device instead of LOCAL_RANK
    from rnascan_amd import cli
    with open(os.path.join(outdir, "out.%d.tsv" % rank), "w") as out:
        cli.main(argv, out=out)                     # real HipEngine, one ctx per process; process group = gloo (the default)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_cli_two_processes_one_gpu(tmp_path):
    """the sharded product path with the REAL engine: two ranks (gloo rendezvous, both on GPU 0 because
    the box has one), each scanning its contiguous share of the records; rank 0 prints the whole table"""
    import socket
    import torch.multiprocessing as mp
    from rnascan_amd import cli
    rng = np.random.default_rng(9)
    fa = tmp_path / "many.fa"
    with open(fa, "w") as f:
        for i in range(200):
            f.write(">rec%d d%d\n%s\n" % (i, i, "".join(rng.choice(list("ACGTN"), size=int(rng.integers(0, 3000)),
                                                                     p=[.245, .245, .245, .245, .02]))))
    argv = ["-p", SEQ_PFM, "-C", "0.01", "-m", "2", str(fa)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_gpu_cli_worker, args=(2, port, str(tmp_path), argv), nprocs=2, join=True)
    single = io.StringIO()
    cli.main(argv, out=single)
    assert open(tmp_path / "out.0.tsv").read() == single.getvalue()
    assert open(tmp_path / "out.1.tsv").read() == ""
    assert single.getvalue().count("\n") > 50


def test_cli_batches_do_not_change_the_table(engine, tmp_path, monkeypatch):
    """RNASCAN_BATCH_POSITIONS small enough to cut a FASTA into many launches: same bytes out"""
    from rnascan_amd import cli
    rng = np.random.default_rng(5)
    fa = tmp_path / "many.fa"
    with open(fa, "w") as f:
        for i in range(60):
            L = int(rng.integers(0, 500))
            f.write(">s%d some text\n%s\n" % (i, "".join(rng.choice(list("ACGTN"), size=L, p=[.24, .24, .24, .24, .04]))))
    argv = ["-p", SEQ_PFM, "-u", "-C", "0.01", "-m", "-3", str(fa)]
    whole = io.StringIO()
    cli.main(argv, engine=engine, out=whole)
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "900")
    cut = io.StringIO()
    cli.main(argv, engine=engine, out=cut)
    assert whole.getvalue().count("\n") > 50
    assert cut.getvalue() == whole.getvalue()


def test_compat_signatures_on_the_gpu(golden):
    """PSSM.search / PSSM.calculate / scan_averaged_structure(struct_file, pssm, minscore) with the reference's
    signatures on the HIP engine, against the reference-generated goldens"""
    from rnascan_amd import scanner
    from test_compat import check_compat
    eng = scanner.HipEngine(0)
    try:
        check_compat(eng, golden)
    finally:
        eng.close()


def test_cli_seq_and_struct_libraries_on_gpu(engine, tmp_path, monkeypatch):
    """config 5's command line -- `rnascan -p seq_library -q struct_library fasta avgdir/` -- through the one-pass
    library kernel: every motif pair of the libraries (18 pairs, mixed widths) on ragged records, the TSV equal to
    the oracle-backed engine's (structure columns within 1e-6), also when cut into several batches"""
    import io
    from engines import OracleEngine
    from rnascan_amd import cli
    from test_scanner_cpu import _library_inputs, assert_tsv_equal
    lib_s, lib_t, fa, d = _library_inputs(tmp_path)
    argv = ["-p", lib_s, "-q", lib_t, "-u", "-C", "0.01", "-m", "-9", "--profile-dtype", "float64", fa, d]
    want = io.StringIO()
    cli.main(argv, engine=OracleEngine(), out=want)
    got = io.StringIO()
    cli.main(argv, engine=engine, out=got)
    assert want.getvalue().count("\n") > 100
    assert_tsv_equal(got.getvalue(), want.getvalue(), tol=1e-6)
    monkeypatch.setenv("RNASCAN_BATCH_POSITIONS", "700")
    cut = io.StringIO()
    cli.main(argv, engine=engine, out=cut)
    assert cut.getvalue() == got.getvalue()
    # a sequence-only library goes through the same kernel without the structure side
    s_want, s_got = io.StringIO(), io.StringIO()
    cli.main(["-p", lib_s, "-u", "-C", "0.01", "-m", "2", fa], engine=OracleEngine(), out=s_want)
    cli.main(["-p", lib_s, "-u", "-C", "0.01", "-m", "2", fa], engine=engine, out=s_got)
    assert s_got.getvalue() == s_want.getvalue() and s_want.getvalue().count("\n") > 50


def test_profile_store_on_gpu(engine, golden, tmp_path):
    """N2 on the device: structure.<id>.txt directory -> packed store (rnascan-pack-profiles) -> scanned as mapped, equal to
    the reference's own scan_main directory-branch output, to the directory scan, and through the command line"""
    import io
    import shutil
    from engines import OracleEngine
    from rnascan_amd import cli, fasta, pssm, scanner, store
    d = tmp_path / "avg"
    d.mkdir()
    shutil.copyfile(os.path.join(DATA_DIR, "HIST2H3C_3p_end_structure.txt"), d / "structure.hg19_dna.txt")
    st = str(tmp_path / "packed")
    assert store.build_store(str(d), st, np.float64) == 1
    P = {"SLBP_struct": pssm.pfm2pssm(STRUCT_PFM, 0.01, fasta.STRUCT, None)}
    ps = store.ProfileStore(st)
    df = scanner.scan_store(engine, ps, P, 0.0, "aligned")
    g = golden["scan_main_dir"]
    got = [[r[0], r[1], r[2], int(r[3]), int(r[4]), r[5], float(r[6])] for r in df.itertuples(index=False)]
    assert list(df.columns) == g["columns"] and len(got) == len(g["rows"])
    for a, b in zip(got, g["rows"]):
        assert a[:6] == b[:6] and abs(a[6] - b[6]) <= 1e-6
    # ragged synthetic directory: store scan (HIP) == directory scan (HIP) == store scan (oracle engine)
    rng = np.random.default_rng(4)
    d2 = tmp_path / "avg2"
    d2.mkdir()
    for i in range(23):
        L = int(rng.integers(0, 400))
        with open(d2 / ("structure.s%02d.txt" % i), "w") as f:
            f.write("PO\t" + "\t".join("BEHLMRT") + "\n")
            p = rng.dirichlet(np.full(7, 0.3), size=L) if L else np.zeros((0, 7))
            for j in range(L):
                f.write(str(j) + "\t" + "\t".join(repr(float(x)) for x in p[j]) + "\n")
    st2 = str(tmp_path / "packed2")
    store.build_store(str(d2), st2, np.float32)
    ps2 = store.ProfileStore(st2)
    a = scanner.scan_store(engine, ps2, P, -40.0, "aligned")
    b = scanner.scan_store(OracleEngine(), ps2, P, -40.0, "aligned")
    assert len(a) == len(b) > 50
    assert a.drop(columns="LogOdds").equals(b.drop(columns="LogOdds")) and np.abs(a["LogOdds"] - b["LogOdds"]).max() <= 1e-6
    named = sorted(scanner.load_profile_dir(str(d2)), key=lambda t: t[0])
    c = scanner.scan_profiles(engine, named, P, -40.0, "aligned", np.float32)
    assert a.drop(columns="LogOdds").equals(c.drop(columns="LogOdds")) and np.abs(a["LogOdds"] - c["LogOdds"]).max() <= 1e-6
    o1, o2 = io.StringIO(), io.StringIO()
    cli.main(["-q", STRUCT_PFM, "-u", "-C", "0.01", "-m", "-40", st2], engine=engine, out=o1)
    cli.main(["-q", STRUCT_PFM, "-u", "-C", "0.01", "-m", "-40", st2], engine=OracleEngine(), out=o2)
    assert_rows = lambda t: [l.split("\t") for l in t.splitlines()]     # noqa: E731
    r1, r2 = assert_rows(o1.getvalue()), assert_rows(o2.getvalue())
    assert len(r1) == len(r2) == len(a) + 1
    for x, y in zip(r1[1:], r2[1:]):
        assert x[:6] == y[:6] and x[7] == y[7] and abs(float(x[6]) - float(y[6])) <= 1e-6


@pytest.mark.parametrize("kind", ["both", "seq", "struct"])
def test_host_pipeline_equals_staged_hits(ctx, oracle, kind):
    """pfmscan_hits_pipeline_host (chunks, upload of chunk k+1 beside the scan of chunk k) == pfmscan_hits_host == oracle,
    with chunk borders inside records and inside windows, for every kind of motif"""
    from test_gpu_parity import rand_stream, rand_struct_pssm, rand_table
    rng = np.random.default_rng({"both": 1, "seq": 2, "struct": 3}[kind])
    s = rand_stream(rng, 150, 0, 2000, foreign=0.003)
    m = 12
    T = rand_table(rng, m, 4) if kind != "struct" else None
    P = rand_struct_pssm(rng, m, inf_frac=0.1) if kind != "seq" else None
    motif = ctx.motif(T, P)
    thr_s, thr_t = (0.0, -30.0)
    want = ctx.hits_host(motif, s.codes if T is not None else None, s.profile if P is not None else None, thr_s, thr_t)
    assert len(want[0]) > 100
    for chunk in (4096, 5120, 65536, 1 << 22):
        got = ctx.hits_pipeline_host(motif, s.codes if T is not None else None, s.profile if P is not None else None, thr_s, thr_t, chunk)
        assert np.array_equal(got[0], want[0])
        if T is not None:
            assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
        if P is not None:
            assert np.abs(got[2] - want[2]).max() <= 1e-6
    # against the oracle (sequence side decides the positions when there is one)
    sq = oracle.stream_seq(s.codes, T) if T is not None else None
    st = oracle.stream_struct(s.profile, P) if P is not None else None
    assert np.array_equal(want[0], oracle.stream_hits(sq, st, thr_s, thr_t))
    from rnascan_amd import _lib
    with pytest.raises(_lib.CapacityError) as ei:
        ctx.hits_pipeline_host(motif, s.codes if T is not None else None, s.profile if P is not None else None, thr_s, thr_t, 4096, capacity=5)
    assert ei.value.required >= len(want[0])
    motif.close()


@pytest.mark.parametrize("minscore", ["8", "-3"])
def test_cli_on_a_nasty_fasta_equals_the_oracle_engine_run(engine, tmp_path, minscore):
    """native ingest + the device scan + native rows against the TEST-ONLY oracle engine behind the same CLI: CRLF,
    wrapped and blank lines, lower case, foreign letters, blanks in the lines, headers that need quoting"""
    from engines import OracleEngine
    from rnascan_amd import cli
    path = str(tmp_path / "n.fa")
    nasty_fasta(path, n=300, seed=11)
    got, want = io.StringIO(), io.StringIO()
    cli.main(["-p", SEQ_PFM, "-u", "-m", minscore, path], engine=engine, out=got)
    cli.main(["-p", SEQ_PFM, "-u", "-m", minscore, path], engine=OracleEngine(), out=want)
    assert got.getvalue() == want.getvalue() and got.getvalue().count("\n") > 100


def test_staged_upload_of_a_mapped_store_equals_the_runtime_copy(ctx, tmp_path, monkeypatch):
    """PFMSCAN_UPLOAD_STAGED (64-MiB pieces through pinned buffers filled by a thread pool; what a mapped profile store
    takes from 256 MB on, forced here for a smaller one) moves the same bytes as the runtime's copy: identical hits
    from pfmscan_stage, the pipeline and hits_host, odd sizes included"""
    from rnascan_amd import _lib
    from test_gpu_parity import rand_struct_pssm, rand_table
    rng = np.random.default_rng(21)
    n_pos = 2_600_017                                        # 72.8 MB of float32 rows: one full piece + a ragged one
    prof = rng.random((n_pos, 7), dtype=np.float32)
    prof /= prof.sum(axis=1, keepdims=True)
    prof[::3001] = 0.0
    codes = rng.integers(0, 4, size=n_pos).astype(np.uint8)
    codes[::3001] = 7
    path = str(tmp_path / "p.f32")
    prof.tofile(path)
    mapped = np.memmap(path, dtype=np.float32, mode="r", shape=(n_pos, 7))
    assert _lib.is_file_mapping(mapped[5:100]) and not _lib.is_file_mapping(prof)
    m = 12

    def same(got, want):                                      # the fused and the two-pass kernels differ in the last bits of a structure score
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
        assert np.abs(got[2] - want[2]).max() <= 1e-9

    motif = ctx.motif(rand_table(rng, m, 4), rand_struct_pssm(rng, m, inf_frac=0.0))
    monkeypatch.setenv("PFMSCAN_UPLOAD", "0")
    want = ctx.hits_host(motif, codes, prof, 2.0, -1e30)
    assert len(want[0]) > 1000
    monkeypatch.setenv("PFMSCAN_UPLOAD", "1")
    for src in (mapped, prof):
        got = ctx.hits_host(motif, codes, src, 2.0, -1e30)
        same(got, want)
        got = ctx.hits_pipeline_host(motif, codes, src, 2.0, -1e30, 1 << 21)
        same(got, want)
    ctx.stage(codes, mapped)
    got = ctx.hits_staged(motif, 2.0, -1e30)
    same(got, want)
    monkeypatch.delenv("PFMSCAN_UPLOAD")
    ctx.stage(codes, mapped)                                  # the mode a mapped source selects by itself (below 256 MB: plain copy)
    assert ctx._upload_mode == 1
    got = ctx.hits_staged(motif, 2.0, -1e30)
    same(got, want)
    ctx.stage(codes, prof)
    assert ctx._upload_mode == 0
    motif.close()


def test_mapped_store_of_300_mb_takes_the_staged_upload_by_itself(ctx, tmp_path, monkeypatch):
    """no environment override: a memory-mapped profile of more than 256 MB selects PFMSCAN_UPLOAD_STAGED in the Python
    layer and goes up through the pinned pieces; the structure scores equal those of the in-memory copy"""
    from test_gpu_parity import rand_struct_pssm
    monkeypatch.delenv("PFMSCAN_UPLOAD", raising=False)
    rng = np.random.default_rng(8)
    n_pos = 10_500_003                                       # 294 MB of float32 rows, a ragged last piece
    prof = rng.random((n_pos, 7), dtype=np.float32)
    path = str(tmp_path / "big.f32")
    prof.tofile(path)
    mapped = np.memmap(path, dtype=np.float32, mode="r", shape=(n_pos, 7))
    motif = ctx.motif(None, rand_struct_pssm(rng, 12, inf_frac=0.0))
    _, want = ctx.scan_host(motif, None, prof, want_seq=False)
    assert ctx._upload_mode == 0
    _, got = ctx.scan_host(motif, None, mapped, want_seq=False)
    assert ctx._upload_mode == 1
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    ctx.stage(None, prof)                                    # and back to the runtime's copy for ordinary memory
    assert ctx._upload_mode == 0
    motif.close()


def test_staged_upload_reads_the_file_behind_a_mapping(ctx, tmp_path, monkeypatch):
    """a numpy.memmap source is registered with the uploader (pfmscan_upload_source_file) and staged transfers pread the
    file instead of touching the mapping: same scores as from memory for a memmap that starts at a file offset, for a
    view into it, and with the pread path switched off; the range is forgotten when the memmap goes away"""
    import gc
    from test_gpu_parity import rand_struct_pssm
    monkeypatch.setenv("PFMSCAN_UPLOAD", "1")                 # staged from 32 MB on
    rng = np.random.default_rng(5)
    n_pos, lead = 1_500_011, 4096 + 28 * 3                    # 42 MB of rows behind a 4180-byte header
    prof = rng.random((n_pos, 7), dtype=np.float32)
    path = str(tmp_path / "with_header.bin")
    with open(path, "wb") as f:
        f.write(b"\x5a" * lead)
        prof.tofile(f)
    mapped = np.memmap(path, dtype=np.float32, mode="r", offset=lead, shape=(n_pos, 7))
    motif = ctx.motif(None, rand_struct_pssm(rng, 12, inf_frac=0.0))
    _, want = ctx.scan_host(motif, None, prof, want_seq=False)
    before = len(ctx.__dict__.get("_mappings", {}))
    _, got = ctx.scan_host(motif, None, mapped, want_seq=False)
    assert len(ctx._mappings) == before + 1                   # registered on first use
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    a, b = 100_003, 1_400_001                                 # a view: the source pointer lies inside the range
    _, got = ctx.scan_host(motif, None, mapped[a:b], want_seq=False)
    _, want_view = ctx.scan_host(motif, None, prof[a:b], want_seq=False)
    assert np.array_equal(got.view(np.uint64), want_view.view(np.uint64))
    monkeypatch.setenv("PFMSCAN_UPLOAD_NO_PREAD", "1")        # the mapping itself as the source: the same bytes
    _, got = ctx.scan_host(motif, None, mapped, want_seq=False)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    monkeypatch.delenv("PFMSCAN_UPLOAD_NO_PREAD")
    del mapped, got
    gc.collect()
    assert len(ctx._mappings) == before                       # forgotten before the mapping went away
    motif.close()


def test_a_repacked_store_is_not_read_in_the_mapping_place(ctx, tmp_path, monkeypatch):
    """the staged uploader preads the FILE a read-only memmap maps -- but only while the path still names that file.  A
    store re-packed by rename after it was opened (same path, other bytes): the registration is refused, the upload reads
    the mapping, the hits are those of the bytes that were mapped."""
    import ctypes
    from rnascan_amd import _lib, store
    from test_gpu_parity import rand_struct_pssm
    rng = np.random.default_rng(4)
    sdir = tmp_path / "st"
    d = tmp_path / "avg"
    d.mkdir()
    L = 3000
    for i in range(4):
        prof = rng.dirichlet(np.full(7, 0.3), size=L)
        with open(d / ("structure.r%d.txt" % i), "w") as g:
            g.write("PO\t" + "\t".join("BEHLMRT") + "\n")
            g.writelines(str(k) + "\t" + "\t".join(str(float(x)) for x in row) + "\n" for k, row in enumerate(prof))
    store.build_store(str(d), str(sdir), np.float32)
    ps = store.ProfileStore(str(sdir))
    mapped = np.array(ps.profile)                             # the bytes that are mapped
    assert ps.profile._mapped_file_id[1] == os.stat(sdir / "profile.f32").st_ino
    # same path, another file (what an atomic re-pack does): other rows
    other = np.ascontiguousarray(mapped[::-1])
    tmp = sdir / "profile.f32.new"
    other.tofile(tmp)
    os.replace(tmp, sdir / "profile.f32")
    L_ = _lib.load()
    rc = L_.pfmscan_upload_source_file_checked(ctx._h, ctypes.c_void_p(ps.profile.ctypes.data), ps.profile.nbytes,
                                               os.fsencode(str(sdir / "profile.f32")), 0, *[int(x) for x in ps.profile._mapped_file_id])
    assert rc == _lib.E_BADARG and b"not the file that was mapped" in L_.pfmscan_last_error(ctx._h)
    motif = ctx.motif(None, rand_struct_pssm(rng, 10, inf_frac=0.0))
    monkeypatch.setenv("PFMSCAN_UPLOAD", "1")                 # force the staged path for this small store
    got = ctx.hits_host(motif, None, ps.profile, -np.inf, -3.0)
    want = ctx.hits_host(motif, None, mapped, -np.inf, -3.0)
    assert len(want[0]) > 100 and np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    motif.close()
