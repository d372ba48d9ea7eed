"""Seeded random configurations through every scan entry point against the oracle: the widths, record shapes, special cells,
profile precisions and thresholds the hand-picked cases of test_gpu_parity / test_gpu_library do not combine.  Each seed
draws ONE configuration (what it drew is in the assertion message); the bar is the usual one -- positions and float32
sequence scores bit-exact (_pwm.c:34-68), structure scores within 1e-6 (rnascan.py:302-307), hit sets exact."""
import numpy as np
import pytest

from conftest import assert_f32_bits_equal, assert_struct_close
from test_gpu_parity import rand_stream, rand_struct_pssm, rand_table

pytestmark = pytest.mark.gpu


def _draw(seed):
    rng = np.random.default_rng(10_000 + seed)
    cfg = {
        "mode": ("seq", "struct", "both")[int(rng.integers(0, 3))],
        "m": int(rng.choice([1, 2, 3, 5, 7, 8, 9, 11, 12, 13, 16, 17, 18, 19, 24, 31, 32, 33, 40, 64])),
        "dtype": (np.float32, np.float64)[int(rng.integers(0, 2))],
        "t_inf": (0.0, 0.1)[int(rng.integers(0, 2))],
        "p_inf": (0.0, 0.08)[int(rng.integers(0, 2))],
        "n_records": int(rng.integers(1, 40)),
        "lo": int(rng.choice([0, 1, 30, 200])),
        "hi": int(rng.choice([40, 300, 2500])),
        "foreign": (0.0, 0.002, 0.05)[int(rng.integers(0, 3))],
        "thr_kind": ("none", "median", "top", "max", "on_a_score")[int(rng.integers(0, 5))],
    }
    cfg["hi"] = max(cfg["hi"], cfg["lo"])
    return rng, cfg


def _threshold(rng, kind, scores):
    scores = np.asarray(scores, dtype=np.float64)
    fin = scores[np.isfinite(scores) & (np.abs(scores) < 1e300)]      # (not the +-DBL_MAX nan_to_num makes of an infinite row-dot)
    if kind == "none" or fin.size == 0:
        return -np.inf
    if kind == "median":
        return float(np.median(fin))
    if kind == "top":
        return float(np.quantile(fin, 0.99))
    if kind == "max":
        return float(fin.max())                        # strict >: no hit
    return float(rng.choice(fin))                      # ON an existing score


@pytest.mark.parametrize("seed", range(240))
def test_random_single_motif_configurations(ctx, oracle, seed):
    rng, cfg = _draw(seed)
    m = cfg["m"]
    s = rand_stream(rng, cfg["n_records"], cfg["lo"], cfg["hi"], foreign=cfg["foreign"], dtype=cfg["dtype"])
    T = rand_table(rng, m, inf_frac=cfg["t_inf"]) if cfg["mode"] != "struct" else None
    P = rand_struct_pssm(rng, m, min(cfg["p_inf"], 1.0 / m)) if cfg["mode"] != "seq" else None
    motif = ctx.motif(T, P) if P is not None else ctx.motif(letter_table=T)
    want_seq = oracle.stream_seq(s.codes, T) if T is not None else None
    want_st = oracle.stream_struct(s.profile, P) if P is not None else None
    got_seq, got_st = ctx.scan_host(motif, s.codes if T is not None else None, s.profile if P is not None else None)
    if T is not None:
        assert_f32_bits_equal(got_seq, want_seq)
    if P is not None:
        assert_struct_close(got_st, want_st)
    thr_seq = _threshold(rng, cfg["thr_kind"], want_seq) if T is not None else -np.inf
    thr_st = _threshold(rng, cfg["thr_kind"], want_st) if P is not None else -np.inf
    pos, sq, st = ctx.hits_host(motif, s.codes if T is not None else None, s.profile if P is not None else None,
                                thr_seq=thr_seq, thr_struct=thr_st)
    want_pos = oracle.stream_hits(want_seq, want_st, thr_seq, thr_st)
    assert np.array_equal(pos, want_pos), (cfg, thr_seq, thr_st, pos.size, want_pos.size)
    if T is not None:
        assert_f32_bits_equal(sq, want_seq[want_pos])
    if P is not None:
        assert_struct_close(st, want_st[want_pos])
    motif.close()


@pytest.mark.parametrize("seed", range(96))
def test_random_library_configurations(ctx, oracle, seed):
    rng, cfg = _draw(500 + seed)
    m = min(cfg["m"], 40)
    n = int(rng.choice([1, 2, 7, 8, 9, 12, 13, 25]))
    s = rand_stream(rng, cfg["n_records"], cfg["lo"], cfg["hi"], foreign=cfg["foreign"], dtype=cfg["dtype"])
    LT = np.stack([rand_table(rng, m, inf_frac=cfg["t_inf"]) for _ in range(n)]) if cfg["mode"] != "struct" else None
    LP = np.stack([rand_struct_pssm(rng, m, min(cfg["p_inf"], 1.0 / m)) for _ in range(n)]) if cfg["mode"] != "seq" else None
    lib = ctx.library(LT, LP)
    want_seq = [oracle.stream_seq(s.codes, LT[k]) for k in range(n)] if LT is not None else None
    want_st = [oracle.stream_struct(s.profile, LP[k]) for k in range(n)] if LP is not None else None
    kind = cfg["thr_kind"] if cfg["thr_kind"] != "none" else "median"          # a library takes finite thresholds
    thr_seq = np.array([_threshold(rng, kind, want_seq[k]) for k in range(n)]) if LT is not None else None
    thr_st = np.array([_threshold(rng, kind, want_st[k]) for k in range(n)]) if LP is not None else None
    if thr_seq is not None:
        thr_seq[~np.isfinite(thr_seq)] = 0.0                                  # (a stream with no scorable window)
    if thr_st is not None:
        thr_st[~np.isfinite(thr_st)] = 0.0
    pos, mot, sq, st = ctx.library_hits_host(lib, s.codes if LT is not None else None, s.profile if LP is not None else None, thr_seq, thr_st)
    for k in range(n):
        sel = mot == k
        want_pos = oracle.stream_hits(want_seq[k] if LT is not None else None, want_st[k] if LP is not None else None,
                                      thr_seq[k] if LT is not None else -np.inf, thr_st[k] if LP is not None else -np.inf)
        assert np.array_equal(pos[sel], want_pos), (cfg, n, k, int(sel.sum()), want_pos.size)
        if LT is not None:
            assert_f32_bits_equal(sq[sel], want_seq[k][want_pos])
        if LP is not None:
            assert_struct_close(st[sel], want_st[k][want_pos])
    lib.close()
