"""Parity of the HIP path (through the C ABI) with the CPU oracle and the
committed goldens.  Integer/position work and float32 sequence scores are
bit-exact; structure scores within 1e-6 absolute (BASELINE.json north_star)."""
import os
import numpy as np
import pytest

from conftest import assert_f32_bits_equal, assert_struct_close

pytestmark = pytest.mark.gpu

from rnascan_amd import pack  # noqa: E402


def rand_table(rng, m, nletters=4, inf_frac=0.0):
    T = np.full((m, 8), np.nan)
    T[:, :nletters] = rng.normal(0, 2, size=(m, nletters))
    if inf_frac:
        mask = rng.random((m, nletters)) < inf_frac
        T[:, :nletters][mask] = -np.inf
    return T


def rand_struct_pssm(rng, m, inf_frac=0.0):
    P = rng.normal(-1, 2.5, size=(m, 7))
    if inf_frac:
        P[rng.random((m, 7)) < inf_frac] = -np.inf
    return P


def rand_stream(rng, n_records, lo, hi, foreign=0.002, zero_snap=True, dtype=np.float32):
    codes, profs = [], []
    for _ in range(n_records):
        L = int(rng.integers(lo, hi + 1))
        c = rng.integers(0, 4, size=L).astype(np.uint8)
        c[rng.random(L) < foreign] = 7
        p = rng.dirichlet(np.full(7, 0.3), size=L) if L else np.zeros((0, 7))
        if zero_snap and L:
            p[p < 0.02] = 0.0
            p /= p.sum(axis=1, keepdims=True)
        codes.append(c)
        profs.append(p.astype(dtype))
    return pack.pack(codes, profs, profile_dtype=dtype)


# ---------------------------------------------------------------------------
# _pwm.calculate drop-in
# ---------------------------------------------------------------------------
def test_pwm_calculate_goldens(ctx, golden):
    for case in golden["pwm"]:
        M = np.array(case["matrix"], dtype=np.float64).reshape(-1, 4)
        got = ctx.pwm_calculate(case["sequence"], M)
        assert_f32_bits_equal(got, np.array(case["scores"], dtype=np.float32))


def test_pwm_calculate_shorter_than_width(ctx, golden):
    assert golden["pwm_shorter_than_m_minus_1_raises"] == "MemoryError"
    with pytest.raises(MemoryError):
        ctx.pwm_calculate("ACG", np.zeros((8, 4)))
    assert ctx.pwm_calculate("ACGUACG", np.zeros((8, 4))).shape == (0,)


def test_pwm_calculate_argument_errors(ctx):
    with pytest.raises(ValueError):
        ctx.pwm_calculate("ACGU", np.zeros((2, 4), dtype=np.float32))      # _pwm.c:96-100
    with pytest.raises(ValueError):
        ctx.pwm_calculate("ACGU", np.zeros((2, 3)))                        # _pwm.c:107-112
    with pytest.raises(ValueError):
        ctx.pwm_calculate("ACGU", np.zeros(4))                             # _pwm.c:101-106
    with pytest.raises(ValueError):
        ctx.pwm_calculate("A" * 5000, np.zeros((4097, 4)))                 # wider than PFMSCAN_MAX_WIDTH


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 8, 12, 15, 16, 17, 18, 31, 32, 33, 48, 64, 65, 100, 257])
def test_pwm_calculate_vs_oracle_widths(ctx, oracle, m):
    rng = np.random.default_rng(1000 + m)
    letters = np.array(list("ACGUacgutTN-"))
    for L in (m, m + 1, m + 3, 255, 1024, 1027, 4096 + 5, 20011):
        seq = "".join(rng.choice(letters, size=L, p=[.2, .2, .2, .2, .03, .03, .03, .03, .03, .02, .02, .01]))
        M = rng.normal(0, 3, size=(m, 4))
        if L < m - 1:                                   # a negative output shape: MemoryError in the reference (_pwm.c:26-31)
            with pytest.raises(MemoryError):
                ctx.pwm_calculate(seq, M)
            continue
        assert_f32_bits_equal(ctx.pwm_calculate(seq, M), oracle.pwm_calculate(seq, M))


def test_calculate_route_golden(ctx, golden):
    g = golden["calculate_route"]
    M = np.array([[g["pssm"][l][i] for l in "ACGU"] for i in range(len(g["pssm"]["A"]))])
    assert_f32_bits_equal(ctx.pwm_calculate(g["sequence"], M), np.array(g["scores"], dtype=np.float32))
    one = ctx.pwm_calculate(g["single_window_sequence"], M)
    assert one.shape == (1,) and float(one[0]) == g["single_window_score"]


# ---------------------------------------------------------------------------
# generic-alphabet letter scan, fp64 out (matrix.py:25-43)
# ---------------------------------------------------------------------------
def test_py_calculate_goldens(ctx, golden):
    for case in golden["py_calculate"]:
        letters = case["letters"]
        T = np.full((len(case["table"]), 8), np.nan)
        T[:, :len(letters)] = np.array(case["table"], dtype=np.float64)
        m = case["m"]
        motif = ctx.motif(letter_table=T[:m])
        codes = pack.pack([pack.encode_letters(case["sequence"], letters)]).codes
        got = ctx.scan_letters_f64_host(motif, codes)[: max(len(case["sequence"]) - m + 1, 0)]
        want = np.array(case["scores"], dtype=np.float64)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])      # fp64 sequential sum: exact
        motif.close()


# ---------------------------------------------------------------------------
# packed-stream scans vs the oracle
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 7, 8, 12, 13, 18, 30, 64])
@pytest.mark.parametrize("inf_frac", [0.0, 0.15])
def test_stream_seqstruct_vs_oracle(ctx, oracle, m, inf_frac):
    rng = np.random.default_rng(7 * m + int(inf_frac * 100))
    s = rand_stream(rng, 40, 0, 700)
    T = rand_table(rng, m, 4, inf_frac=inf_frac / 3)
    P = rand_struct_pssm(rng, m, inf_frac=inf_frac)
    motif = ctx.motif(T, P)
    got_seq, got_st = ctx.scan_host(motif, s.codes, s.profile)
    assert_f32_bits_equal(got_seq, oracle.stream_seq(s.codes, T))
    assert_struct_close(got_st, oracle.stream_struct(s.profile, P))
    motif.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_pos_extra", [0, 1, 2, 3, 5, 1791, 1792, 1793])
def test_stream_tail_sizes(ctx, oracle, dtype, n_pos_extra):
    """stream lengths around the tile sizes (1024 / 1280 / 1792 / 4096) and tiny streams"""
    rng = np.random.default_rng(99 + n_pos_extra)
    m = 12
    T = rand_table(rng, m)
    P = rand_struct_pssm(rng, m, inf_frac=0.1)
    motif = ctx.motif(T, P)
    for base in (0, 11, 12, 13, 4096):
        L = base + n_pos_extra
        if L == 0:
            continue
        s = rand_stream(rng, 1, L, L, dtype=dtype)
        got_seq, got_st = ctx.scan_host(motif, s.codes, s.profile)
        assert_f32_bits_equal(got_seq, oracle.stream_seq(s.codes, T))
        assert_struct_close(got_st, oracle.stream_struct(s.profile, P))
    motif.close()


def test_stream_struct_only_and_seq_only(ctx, oracle):
    rng = np.random.default_rng(5)
    m = 12
    s = rand_stream(rng, 25, 50, 900)
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m)
    both = ctx.motif(T, P)
    only_seq = ctx.motif(letter_table=T)
    only_st = ctx.motif(struct_pssm=P)
    want_seq, want_st = oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P)
    a, b = ctx.scan_host(only_seq, s.codes)
    assert b is None
    assert_f32_bits_equal(a, want_seq)
    a, b = ctx.scan_host(only_st, None, s.profile)
    assert a is None
    assert_struct_close(b, want_st)
    a, b = ctx.scan_host(both, s.codes, s.profile, want_struct=False)
    assert_f32_bits_equal(a, want_seq)
    a, b = ctx.scan_host(both, s.codes, s.profile, want_seq=False)
    assert_struct_close(b, want_st)


def test_profile_with_nonfinite_values_takes_exact_path(ctx, oracle):
    """finite PSSM -> fast kernel; NaN/inf planted in the PROFILE must still follow
    nan_to_num per row-dot (rnascan.py:306)"""
    rng = np.random.default_rng(17)
    m = 12
    s = rand_stream(rng, 6, 300, 600)
    idx = rng.integers(0, s.n_pos, size=25)
    s.profile[idx[:10], rng.integers(0, 7, size=10)] = np.nan
    s.profile[idx[10:18], rng.integers(0, 7, size=8)] = np.inf
    s.profile[idx[18:], rng.integers(0, 7, size=7)] = -np.inf
    P = rand_struct_pssm(rng, m)
    motif = ctx.motif(struct_pssm=P)
    _, got = ctx.scan_host(motif, None, s.profile)
    assert_struct_close(got, oracle.stream_struct(s.profile, P))


def test_scan_averaged_structure_goldens(ctx, golden, data_dir):
    """the reference's own scan_averaged_structure output (rnascan.py:293-315)"""
    import os
    for case in golden["scan_averaged_structure"]:
        if case["profile_file"]:
            rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(data_dir, case["profile_file"]))][1:]
            prof = np.array([[float(x) for x in r[1:]] for r in rows])
        else:
            prof = np.array(case["profile"], dtype=np.float64)
        P = np.array(case["pssm"], dtype=np.float64)
        m = P.shape[0]
        motif = ctx.motif(struct_pssm=P)
        s = pack.pack(profiles=[prof], profile_dtype=np.float64)
        _, got = ctx.scan_host(motif, None, s.profile)
        got = got[: prof.shape[0] - m + 1]
        keep = got > case["minscore"]                       # rnascan.py:310, strict
        starts = np.flatnonzero(keep) + 1                   # 1-based
        want = case["rows"]
        assert [r[0] for r in want] == starts.tolist()
        assert [r[1] for r in want] == (starts + m - 1).tolist()
        assert_struct_close(got[keep], np.array([r[2] for r in want]))
        # float32-stored profile (the headline storage): same hits at the reference's numbers
        s32 = pack.pack(profiles=[prof], profile_dtype=np.float32)
        _, got32 = ctx.scan_host(motif, None, s32.profile)
        got32 = got32[: prof.shape[0] - m + 1]
        # north_star's 1e-6: float32 storage of the reference's own example stays inside it (8.6e-7 at |score| <= 37); it
        # leaves the tolerance for wider PFMs / larger |log-odds|, where --profile-dtype float64 is the answer
        assert_struct_close(got32[keep], np.array([r[2] for r in want]), tol=1e-6)
        motif.close()


# ---------------------------------------------------------------------------
# hits mode
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("thr", [-np.inf, -5.0, 0.0, 6.0])
def test_hits_vs_oracle(ctx, oracle, thr):
    rng = np.random.default_rng(31)
    m = 8
    s = rand_stream(rng, 60, 10, 1500)
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m) * 0.3 + 1.0
    both = ctx.motif(T, P)
    want_seq, want_st = oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P)
    pos, sq, st = ctx.hits_host(both, s.codes, s.profile, thr_seq=thr, thr_struct=thr)
    want_pos = oracle.stream_hits(want_seq, want_st, thr, thr)
    assert np.array_equal(pos, want_pos)                    # integer positions: exact
    assert_f32_bits_equal(sq, want_seq[want_pos])
    assert_struct_close(st, want_st[want_pos])
    # sequence-only hits (config 2 hits mode)
    only = ctx.motif(letter_table=T)
    pos, sq, st = ctx.hits_host(only, s.codes, thr_seq=thr)
    want_pos = oracle.stream_hits(want_seq, None, thr, thr)
    assert np.array_equal(pos, want_pos) and st is None
    assert_f32_bits_equal(sq, want_seq[want_pos])


def test_hits_capacity_protocol(ctx, oracle):
    from rnascan_amd import _lib
    rng = np.random.default_rng(3)
    s = rand_stream(rng, 10, 500, 800)
    T = rand_table(rng, 8)
    motif = ctx.motif(letter_table=T)
    want = oracle.stream_hits(oracle.stream_seq(s.codes, T), None, 0.0, 0.0)
    with pytest.raises(_lib.CapacityError) as e:
        ctx.hits_host(motif, s.codes, thr_seq=0.0, capacity=3)
    assert e.value.required >= len(want)            # a capacity that is guaranteed to suffice
    pos, _, _ = ctx.hits_host(motif, s.codes, thr_seq=0.0, capacity=len(want))
    assert np.array_equal(pos, want)


def test_no_window_spans_two_records(ctx):
    """separator semantics: every window touching a record end scores NaN"""
    rng = np.random.default_rng(8)
    m = 12
    s = rand_stream(rng, 30, 0, 40, foreign=0.0)
    T = rand_table(rng, m)
    motif = ctx.motif(letter_table=T)
    got, _ = ctx.scan_host(motif, s.codes)
    mask = s.window_mask(m)
    assert np.isnan(got[~mask]).all()
    assert not np.isnan(got[mask]).any()


def test_c_program_through_the_abi(tmp_path):
    """a plain-C program linked against libpfmscan.so (no Python, no torch in the process)"""
    import os
    import subprocess
    from conftest import REPO
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(REPO, "rnascan_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "c", "abi_smoke.c"), "-o", exe,
                           "-L", libdir, "-l:libpfmscan.so", "-lm", "-Wl,-rpath," + libdir + ":/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK")


# ---------------------------------------------------------------------------
# hits over a 4-letter alphabet: the fp32 two-letter prefilter (k_letters_pre) must lose no hit
# and report only exact scores
# ---------------------------------------------------------------------------
def _seq_hits_want(oracle, s, T, thr):
    want_seq = oracle.stream_seq(s.codes, T)
    return want_seq, oracle.stream_hits(want_seq, None, thr, thr)


@pytest.mark.parametrize("m", [1, 2, 3, 7, 8, 15, 16, 17, 32, 33, 64])
def test_prefilter_thresholds_on_existing_scores(ctx, oracle, m):
    """thresholds set ON scores that occur (strict >: that score is out) and one float32 below (in)"""
    rng = np.random.default_rng(1000 + m)
    s = rand_stream(rng, 40, 0, 900, foreign=0.01)
    T = rand_table(rng, m)
    motif = ctx.motif(letter_table=T)
    want_seq = oracle.stream_seq(s.codes, T)
    finite = np.unique(want_seq[np.isfinite(want_seq)])
    assert finite.size >= min(4 ** m, 10)
    picks = np.unique(finite[np.linspace(finite.size * 0.6, finite.size - 1, 6).astype(int)])
    for v in picks:
        for thr in (float(v), float(np.nextafter(np.float32(v), np.float32(-np.inf)))):
            pos, sq, _ = ctx.hits_host(motif, s.codes, thr_seq=thr)
            want_pos = oracle.stream_hits(want_seq, None, thr, thr)
            assert np.array_equal(pos, want_pos), (m, thr)
            assert_f32_bits_equal(sq, want_seq[want_pos])
    motif.close()


def test_prefilter_foreign_letters_alias_to_high_scores(ctx, oracle):
    """N / separator codes look like U (low two bits) to the prefilter: the exact pass must drop them"""
    rng = np.random.default_rng(5)
    m = 6
    T = np.full((m, 8), np.nan)
    T[:, :4] = -3.0
    T[:, 3] = 2.0                                    # all-U scores 12
    codes = []
    for _ in range(50):
        c = np.full(int(rng.integers(m, 60)), 3, dtype=np.uint8)
        c[rng.integers(0, c.size, size=2)] = rng.choice([4, 5, 6, 7], size=2)
        codes.append(c)
    s = pack.pack(codes)
    motif = ctx.motif(letter_table=T)
    want_seq, want_pos = _seq_hits_want(oracle, s, T, 11.0)
    pos, sq, _ = ctx.hits_host(motif, s.codes, thr_seq=11.0)
    assert want_pos.size > 0 and np.array_equal(pos, want_pos)
    assert_f32_bits_equal(sq, want_seq[want_pos])
    motif.close()


@pytest.mark.parametrize("kind", ["neg_inf", "pos_inf", "nan_cell", "huge", "beyond_fp32", "tiny"])
def test_prefilter_special_table_values(ctx, oracle, kind):
    rng = np.random.default_rng(77)
    m = 9
    T = rand_table(rng, m)
    if kind == "neg_inf":
        T[:, :4][rng.random((m, 4)) < 0.2] = -np.inf
    elif kind == "pos_inf":
        T[2, 1] = np.inf
        T[5, 0] = -np.inf                            # +inf and -inf in one window: NaN, never a hit
    elif kind == "nan_cell":
        T[4, 2] = np.nan
    elif kind == "huge":
        T[:, :4] *= 1e28                             # inside fp32 range: prefilter stays on, with a wide margin
    elif kind == "beyond_fp32":
        T[:, :4] *= 1e36                             # would overflow fp32: the table is not built, exact kernel runs
    elif kind == "tiny":
        T[:, :4] *= 1e-42                            # fp32 subnormals
    s = rand_stream(rng, 30, 0, 700, foreign=0.01)
    motif = ctx.motif(letter_table=T)
    want_seq = oracle.stream_seq(s.codes, T)
    fin = want_seq[np.isfinite(want_seq)]
    for thr in (-np.inf, float(np.median(fin)), float(np.quantile(fin, 0.99)), float(fin.max())):
        pos, sq, _ = ctx.hits_host(motif, s.codes, thr_seq=thr)
        want_pos = oracle.stream_hits(want_seq, None, thr, thr)
        assert np.array_equal(pos, want_pos), (kind, thr)
        assert_f32_bits_equal(sq, want_seq[want_pos])
    motif.close()


def test_prefilter_dense_hits_overflow_the_lds_queue(ctx, oracle):
    """every window a hit: tiles denser than the 1024-entry queue go straight to global"""
    rng = np.random.default_rng(9)
    m = 5
    T = rand_table(rng, m)
    s = rand_stream(rng, 8, 30000, 40000, foreign=0.0)
    motif = ctx.motif(letter_table=T)
    for thr in (-1e9, -2.0):
        want_seq, want_pos = _seq_hits_want(oracle, s, T, thr)
        pos, sq, _ = ctx.hits_host(motif, s.codes, thr_seq=thr, capacity=s.codes.size)
        assert np.array_equal(pos, want_pos)
        assert_f32_bits_equal(sq, want_seq[want_pos])
    motif.close()


def test_prefilter_off_gives_the_same_hits(ctx, oracle, monkeypatch):
    """PFMSCAN_PREFILTER=0 (the exact kernel only) and the default path agree"""
    from rnascan_amd import _lib
    rng = np.random.default_rng(12)
    T = rand_table(rng, 8)
    s = rand_stream(rng, 50, 100, 3000)
    motif = ctx.motif(letter_table=T)
    pos1, sq1, _ = ctx.hits_host(motif, s.codes, thr_seq=3.0)
    monkeypatch.setenv("PFMSCAN_PREFILTER", "0")
    plain = _lib.Context(0)
    m2 = plain.motif(letter_table=T)
    pos0, sq0, _ = plain.hits_host(m2, s.codes, thr_seq=3.0)
    assert np.array_equal(pos0, pos1)
    assert_f32_bits_equal(sq0, sq1)
    m2.close()
    plain.close()
    motif.close()


@pytest.mark.parametrize("tpb", [2, 3, 7])
def test_prefilter_multi_tile_walk(oracle, monkeypatch, tpb):
    """a workgroup walking several tiles (double-buffered codes, queue flushes at tile boundaries and in
    mid-tile at dense thresholds); forced on a small stream with PFMSCAN_TILES_PER_BLOCK"""
    from rnascan_amd import _lib
    monkeypatch.setenv("PFMSCAN_TILES_PER_BLOCK", str(tpb))
    c = _lib.Context(0)
    rng = np.random.default_rng(40 + tpb)
    m = 10
    T = rand_table(rng, m)
    s = rand_stream(rng, 25, 500, 4000, foreign=0.003)          # ~ 14 tiles of 4096 positions, ragged tail
    motif = c.motif(letter_table=T)
    want_seq = oracle.stream_seq(s.codes, T)
    fin = want_seq[np.isfinite(want_seq)]
    for thr in (float(np.quantile(fin, 0.999)), float(np.quantile(fin, 0.9)), float(np.quantile(fin, 0.3)), -1e30):
        want_pos = oracle.stream_hits(want_seq, None, thr, thr)
        pos, sq, _ = c.hits_host(motif, s.codes, thr_seq=thr)      # (few workgroups -> skewed shards: let it grow)
        assert np.array_equal(pos, want_pos), (tpb, thr)
        assert_f32_bits_equal(sq, want_seq[want_pos])
    motif.close()
    c.close()


def test_host_hits_come_back_sorted_at_millions_of_hits(ctx, oracle):
    """the device-side pack + radix sort + gather of the sharded hit buffers (pfmscan_sort.hip) on a stream
    long enough for several sort passes and many workgroups per shard; seq-only and seq + struct"""
    rng = np.random.default_rng(2024)
    m = 6
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m) * 0.2 + 0.5
    s = rand_stream(rng, 1500, 2000, 3500, foreign=0.001)
    want_seq, want_st = oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P)
    only = ctx.motif(letter_table=T)
    pos, sq, _ = ctx.hits_host(only, s.codes, thr_seq=-0.5)
    want = oracle.stream_hits(want_seq, None, -0.5, -0.5)
    assert want.size > 1_000_000
    assert np.array_equal(pos, want)
    assert_f32_bits_equal(sq, want_seq[want])
    only.close()
    both = ctx.motif(T, P)
    for thr in (-1.0, 1.5):                                  # fused pass (not selective) / candidate-then-verify
        pos, sq, st = ctx.hits_host(both, s.codes, s.profile, thr_seq=thr, thr_struct=-3.0)
        want = oracle.stream_hits(want_seq, want_st, thr, -3.0)
        assert want.size > 10_000
        assert np.array_equal(pos, want)
        assert_f32_bits_equal(sq, want_seq[want])
        assert_struct_close(st, want_st[want])
    both.close()


def test_integration_md_pwm_stub_runs_verbatim(tmp_path, golden):
    """the `_pwm.py` drop-in printed in INTEGRATION.md section 2, executed as written (only the library path
    filled in) in a fresh interpreter without torch: same float32 scores as the goldens, same exceptions"""
    import json
    import os
    import re
    import subprocess
    import sys
    from conftest import REPO
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# rnascan/BioAddons/motifs/_pwm\.py.*?)```", text, re.S).group(1)
    lib = os.path.join(REPO, "rnascan_amd", "libpfmscan.so")
    assert "/path/to/rnascan_amd/libpfmscan.so" in block
    (tmp_path / "_pwm.py").write_text(block.replace("/path/to/rnascan_amd/libpfmscan.so", lib))
    cases = [{"sequence": c["sequence"], "matrix": c["matrix"]} for c in golden["pwm"][:6]]
    (tmp_path / "cases.json").write_text(json.dumps(cases))
    driver = (
        "import json, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import _pwm\n"
        "out = []\n"
        "for c in json.load(open(%r)):\n"
        "    M = np.array(c['matrix'], dtype=np.float64).reshape(-1, 4)\n"
        "    out.append(_pwm.calculate(c['sequence'], M).view(np.uint32).tolist())\n"
        "try:\n"
        "    _pwm.calculate('ACGU', np.zeros((2, 3)))\n"
        "    out.append('no error')\n"
        "except ValueError as e:\n"
        "    out.append('ValueError')\n"
        "print(json.dumps(out))\n" % (str(tmp_path), str(tmp_path / "cases.json")))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    run = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    got = json.loads(run.stdout.strip().splitlines()[-1])
    assert got[-1] == "ValueError"
    for bits, case in zip(got[:-1], golden["pwm"][:6]):
        want = np.array(case["scores"], dtype=np.float32).view(np.uint32)
        have = np.array(bits, dtype=np.uint32)
        nan = np.isnan(want.view(np.float32))
        assert np.array_equal(np.isnan(have.view(np.float32)), nan) and np.array_equal(have[~nan], want[~nan])


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13, 15, 16, 17, 18, 20, 21, 24, 25, 28, 29, 31, 32, 33])
def test_integer_prefilter_kernel_every_width(ctx, oracle, monkeypatch, m):
    """k_letters_cred (position-keyed integer credits, PFMs up to width 32; 33 takes the fp32 prefilter) against the oracle and against the fp32
    prefilter kernel (PFMSCAN_CREDITS=0), thresholds from 'nothing passes' to 'everything passes', ON scores included,
    -inf cells, foreign letters, several tiles per workgroup"""
    from rnascan_amd import _lib
    rng = np.random.default_rng(900 + m)
    T = rand_table(rng, m, 4, inf_frac=min(0.1, 0.6 / m) if m % 3 == 0 else 0.0)     # wide PFMs: keep some windows finite
    s = rand_stream(rng, 30, 0, 2500, foreign=0.004)
    want_seq = oracle.stream_seq(s.codes, T)
    fin = np.sort(want_seq[np.isfinite(want_seq)].astype(np.float64))
    thrs = [1e30, float(fin[-1]), float(fin[int(0.999 * (fin.size - 1))]), float(fin[int(0.97 * (fin.size - 1))]),
            float(np.nextafter(np.float32(fin[int(0.97 * (fin.size - 1))]), np.float32(-np.inf))), 6.0, 0.0, float(fin[0]), -1e30]
    monkeypatch.setenv("PFMSCAN_TILES_PER_BLOCK", "3")
    cred = _lib.Context(0)                                # default: two-letter credit tables keyed by position (k_letters_cred)
    monkeypatch.setenv("PFMSCAN_QUAD", "1")
    quad = _lib.Context(0)                                # four-letter credit tables (k_letters_quad; the A/B variant)
    monkeypatch.setenv("PFMSCAN_QUAD", "0")
    monkeypatch.setenv("PFMSCAN_CREDITS", "0")
    plain = _lib.Context(0)                               # fp32 prefilter (k_letters_pre)
    m2, m1, m0 = quad.motif(letter_table=T), cred.motif(letter_table=T), plain.motif(letter_table=T)
    for thr in thrs + thrs[2:4]:                          # (thresholds come back: the cached table is rebuilt)
        want_pos = oracle.stream_hits(want_seq, None, thr, thr)
        for c, mo in ((quad, m2), (cred, m1), (plain, m0)):
            pos, sq, _ = c.hits_host(mo, s.codes, thr_seq=thr)
            assert np.array_equal(pos, want_pos), (m, thr)
            assert_f32_bits_equal(sq, want_seq[want_pos])
    for mo in (m2, m1, m0):
        mo.close()
    for c in (quad, cred, plain):
        c.close()


def test_staged_length_follows_the_library_not_a_stale_copy(ctx, oracle):
    """scan_host / hits_host / pwm_calculate restage inside the library: a following scan_staged sizes its outputs from
    pfmscan_staged_positions, never from the length of an earlier stage() (that was a host heap overflow)"""
    rng = np.random.default_rng(17)
    small = rand_stream(rng, 3, 50, 80)
    large = rand_stream(rng, 40, 200, 900)
    T = rand_table(rng, 8)
    motif = ctx.motif(letter_table=T)
    ctx.stage(small.codes)
    ctx.scan_host(motif, large.codes)                    # restages the longer stream
    sq, _ = ctx.scan_staged(motif)
    assert sq.shape[0] == large.n_pos
    assert_f32_bits_equal(sq, oracle.stream_seq(large.codes, T))
    ctx.pwm_calculate("ACGUACGUACGUACGU", np.zeros((8, 4)))     # restages 16 positions
    assert ctx.scan_staged(motif)[0].shape[0] == 16
    ctx.hits_pipeline_host(motif, large.codes, None, 2.0, -np.inf, 4096)      # leaves nothing staged
    with pytest.raises(ValueError):
        ctx.scan_staged(motif)
    motif.close()


FIXED_WIDTHS = list(range(4, 19))          # widths with an unrolled instantiation of k_profile (pfmscan_profile_fixed.hip: launch_profile_fixed)


@pytest.mark.parametrize("m", FIXED_WIDTHS)
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("has_seq", [True, False])
@pytest.mark.parametrize("inf_frac", [0.0, 0.1])
def test_fixed_width_profile_kernel_equals_the_generic_one(ctx, oracle, monkeypatch, m, dtype, has_seq, inf_frac):
    """k_profile_fixed (the PFM width a compile-time constant: straight-line row loop, loads requested a step ahead) performs
    the generic kernel's operations in the generic kernel's order: the same bits, structure scores included, on a stream of
    several tiles with a ragged end, NaN / inf cells in the PSSM (the per-row nan_to_num variant) and in the profile (the exact
    re-run of a window whose fast sum came out non-finite)"""
    rng = np.random.default_rng(1000 * m + 10 * int(has_seq) + int(inf_frac * 10) + (dtype == np.float64))
    s = rand_stream(rng, 9, 0, 1500, dtype=dtype)
    bad = rng.integers(0, s.n_pos, size=12)
    s.profile[bad[:6], rng.integers(0, 7, size=6)] = np.nan
    s.profile[bad[6:], rng.integers(0, 7, size=6)] = np.inf
    T = rand_table(rng, m, 4, inf_frac=inf_frac / 3) if has_seq else None
    P = rand_struct_pssm(rng, m, inf_frac=inf_frac)
    motif = ctx.motif(T, P)
    codes = s.codes if has_seq else None
    monkeypatch.setenv("PFMSCAN_PROFILE_FIXED_MIN", "0")          # every instantiation, also the widths the launcher leaves to the generic kernel
    fixed_seq, fixed_st = ctx.scan_host(motif, codes, s.profile)
    monkeypatch.setenv("PFMSCAN_PROFILE_GENERIC", "1")
    gen_seq, gen_st = ctx.scan_host(motif, codes, s.profile)
    # the fused hits pass of the same instantiation (seq > thr && struct > thr; half of the windows pass the letter threshold,
    # so the combined scan takes the fused k_profile pass, not letters first): the same hits with the same score bits
    fin = gen_st[np.isfinite(gen_st)]
    thr_t = float(np.quantile(fin, 0.5)) if len(fin) else 0.0
    thr_s = -np.inf
    if has_seq:
        fs = gen_seq[np.isfinite(gen_seq)].astype(np.float64)
        thr_s = float(np.quantile(fs, 0.3)) if len(fs) else 0.0
    gen_hits = ctx.hits_host(motif, codes, s.profile, thr_s, thr_t)
    monkeypatch.delenv("PFMSCAN_PROFILE_GENERIC")
    fixed_hits = ctx.hits_host(motif, codes, s.profile, thr_s, thr_t)
    assert (len(gen_hits[0]) > 20 or inf_frac > 0) and np.array_equal(fixed_hits[0], gen_hits[0])
    assert np.array_equal(fixed_hits[2].view(np.uint64), gen_hits[2].view(np.uint64))
    if has_seq:
        assert np.array_equal(fixed_hits[1].view(np.uint32), gen_hits[1].view(np.uint32))
    motif.close()
    assert np.array_equal(fixed_st.view(np.uint64), gen_st.view(np.uint64))
    assert_struct_close(fixed_st, oracle.stream_struct(s.profile, P))
    if has_seq:
        assert np.array_equal(fixed_seq.view(np.uint32), gen_seq.view(np.uint32))
        assert_f32_bits_equal(fixed_seq, oracle.stream_seq(s.codes, T))
    else:
        assert fixed_seq is None and gen_seq is None


def test_placed_arrays_hold_what_is_written_and_scan_like_any_other(ctx, oracle):
    """pfmscan_place_alloc (include/pfmscan.h): sets of device arrays in chunks of physical memory chosen by measurement.  The
    arrays are ordinary device memory: what is written is read back, a scan on them gives the oracle's scores, and -- the hazard
    the module's note on address ranges is about -- sets allocated AFTER a set was freed are intact too (on ROCm 7.2 a reused
    address range read back other bytes; the ranges are therefore never reused)."""
    import torch
    from rnascan_amd import _lib
    rng = np.random.default_rng(5)
    m = 12
    s = rand_stream(rng, 30, 200, 900)
    T, P = rand_table(rng, m), rand_struct_pssm(rng, m)
    motif = ctx.motif(T, P)
    want_seq, want_st = oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P)
    dev = torch.device("cuda:0")
    n = s.n_pos
    big = 96 << 20                                  # every array spans several 64 MB chunks (PFMSCAN_PLACE_CHUNK_MB below)
    os.environ["PFMSCAN_PLACE_CHUNK_MB"] = "64"
    try:
        first_ptrs = None
        for cycle, plain in enumerate((False, True, False, False)):
            roomy = torch.cuda.mem_get_info()[0] > (110 << 30)       # the second attempt holds 48 GB and wants 96 GB spare
            if cycle == 3:
                os.environ["PFMSCAN_PLACE_FORCE_RETRY"] = "1"      # the path taken when no pair of candidates is independent
                big += 64 << 20                                     # other sizes: a new set, not the retired one of cycle 2
            arrs = ctx.place_alloc([big + n * 28, big + n * 8, big + n * 4, big + n], plain=plain)
            os.environ.pop("PFMSCAN_PLACE_FORCE_RETRY", None)
            note = ctx.place_note()
            assert ("NOT tuned" in note) == plain, note
            assert ("second attempt" in note) == (cycle == 3 and roomy), note
            # a freed set is retired, not unmapped: the same request gets the same arrays back (no new address range)
            if cycle == 0:
                first_ptrs = [a.ptr for a in arrs]
            assert ("handed out again" in note) == (cycle == 2), note
            assert ([a.ptr for a in arrs] == first_ptrs) == (cycle in (0, 2))
            raw = [torch.as_tensor(a, device=dev) for a in arrs]
            assert all(t.data_ptr() == a.ptr and t.numel() == a.nbytes for t, a in zip(raw, arrs))
            pattern = [torch.randint(0, 255, (a.nbytes,), dtype=torch.uint8, device=dev) for a in arrs]
            for t, p in zip(raw, pattern):
                t.copy_(p)
            torch.cuda.synchronize()
            for t, p in zip(raw, pattern):
                assert torch.equal(t, p), "cycle %d: a placed array does not hold what was written" % cycle
            off = big                                # the scan's arrays at the END of the placed ranges: the last chunks are used too
            raw[0][off:off + n * 28].copy_(torch.from_numpy(s.profile.view(np.uint8).reshape(-1)).to(dev))
            raw[3][off:off + n].copy_(torch.from_numpy(s.codes).to(dev))
            torch.cuda.synchronize()
            ctx.scan_dev(motif, arrs[3].ptr + off, arrs[0].ptr + off, _lib.PROFILE_F32, n, arrs[2].ptr + off, arrs[1].ptr + off, None)
            ctx.synchronize()
            got_seq = raw[2][off:off + n * 4].view(torch.float32).cpu().numpy()
            got_st = raw[1][off:off + n * 8].view(torch.float64).cpu().numpy()
            assert_f32_bits_equal(got_seq, want_seq)
            assert_struct_close(got_st, want_st)
            del raw, pattern
            ctx.place_free(arrs[0])
        with pytest.raises(Exception):
            ctx.place_free(arrs[0])                  # freed already
        ctx.place_trim()                             # the retired sets give their memory back; a new request measures again
        arrs = ctx.place_alloc([big + n * 28, big + n * 8, big + n * 4, big + n])
        assert "handed out again" not in ctx.place_note()
        ctx.place_free(arrs[0])
        # the address-space budget: ranges are never given back, so a context that keeps asking for NEW sizes is told to stop
        os.environ["PFMSCAN_PLACE_VA_BUDGET_GB"] = "1"
        with pytest.raises(MemoryError):
            ctx.place_alloc([big + (1 << 20), big, big, big], plain=True)
        os.environ.pop("PFMSCAN_PLACE_VA_BUDGET_GB", None)
        with pytest.raises(Exception):
            ctx.place_alloc([0, 16])
    finally:
        del os.environ["PFMSCAN_PLACE_CHUNK_MB"]
    motif.close()


@pytest.mark.parametrize("m", list(range(2, 33)))
def test_fixed_width_letters_kernel_equals_the_generic_one(ctx, oracle, monkeypatch, m):
    """k_letters_fixed (all float32 scores, widths 2..32: four windows per lane, rows 0 + 1 as one pair look-up built in the
    reference's order) gives the width-generic k_letters' bits -- and the oracle's -- on ragged records with foreign letters,
    -inf cells, 8-letter codes and every tail length of the last tile"""
    rng = np.random.default_rng(4000 + m)
    T = rand_table(rng, m, nletters=4 if m % 2 else 7, inf_frac=0.1 if m % 3 == 0 else 0.0)
    motif = ctx.motif(letter_table=T)
    for extra in (0, 1, 2, 3, 5, 4095, 4096, 4097):
        lengths = [3000, 0, m - 1, m, m + 1, 777] + ([extra] if extra else [])
        codes = []
        for L in lengths:
            c = rng.integers(0, 7 if m % 2 == 0 else 4, size=L).astype(np.uint8)
            c[rng.random(L) < 0.01] = 7
            codes.append(c)
        s = pack.pack(codes)
        want = oracle.stream_seq(s.codes, T)
        got, _ = ctx.scan_host(motif, s.codes)
        monkeypatch.setenv("PFMSCAN_LETTERS_GENERIC", "1")
        gen, _ = ctx.scan_host(motif, s.codes)
        monkeypatch.delenv("PFMSCAN_LETTERS_GENERIC")
        keep = s.window_mask(m)
        assert_f32_bits_equal(got[keep], want[keep])
        assert np.array_equal(got.view(np.uint32)[~np.isnan(got)], gen.view(np.uint32)[~np.isnan(gen)]) and np.array_equal(np.isnan(got), np.isnan(gen))
    motif.close()


def test_quad_tables_by_threshold_across_streams(oracle, monkeypatch):
    """k_letters_quad (PFMSCAN_QUAD=1: the four-letter credit tables, an A/B path) keeps its credit tables per threshold in a ring of four slots, uploaded asynchronously on the caller's
    stream (no device-wide wait in the `_dev` entry point): seven thresholds cycled through two streams -- slot reuse,
    a slot filled on one stream and read from the other -- give the oracle's hits every time"""
    import torch
    from rnascan_amd import _lib
    monkeypatch.setenv("PFMSCAN_QUAD", "1")
    ctx = _lib.Context(0)
    rng = np.random.default_rng(2024)
    m = 8
    s = rand_stream(rng, 300, 500, 3000)
    T = rand_table(rng, m)
    motif = ctx.motif(letter_table=T)
    want_seq = oracle.stream_seq(s.codes, T)
    fin = want_seq[np.isfinite(want_seq)].astype(np.float64)
    thrs = [float(np.quantile(fin, q)) for q in (0.999, 0.99, 0.995, 0.9995, 0.98, 0.997, 0.9999)]
    dev = torch.device("cuda", 0)
    d_codes = torch.from_numpy(s.codes).to(dev)
    n = int(s.codes.size)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    cap = 1 << 18
    runs = []
    torch.cuda.synchronize()
    for rep in range(3):
        for k, thr in enumerate(thrs + thrs[:2]):                 # 9 calls per round: the ring of four turns over twice
            st = streams[(k + rep) % 2]
            hp = torch.empty(cap, dtype=torch.int64, device=dev)
            hs = torch.empty(cap, dtype=torch.float32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int64, device=dev)
            st.wait_stream(torch.cuda.current_stream())
            ctx.hits_dev(motif, d_codes.data_ptr(), None, _lib.PROFILE_NONE, n, thr, -np.inf, cap, hp.data_ptr(), hs.data_ptr(), None,
                         cnt.data_ptr(), stream=st.cuda_stream)
            runs.append((thr, hp, hs, cnt))
    torch.cuda.synchronize()
    for thr, hp, hs, cnt in runs:
        k = int(cnt.item())
        pos, sc = hp[:k].cpu().numpy(), hs[:k].cpu().numpy()
        order = np.argsort(pos, kind="stable")
        want_pos = oracle.stream_hits(want_seq, None, thr, -np.inf)
        assert np.array_equal(pos[order], want_pos), (thr, k, want_pos.size)
        assert_f32_bits_equal(sc[order], want_seq[want_pos])
    motif.close()
    ctx.close()
