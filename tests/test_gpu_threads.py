"""Two host threads, each with its OWN context (and motifs) on the same device, scanning at the same time: the boundary's
threading contract (thread-safe per ctx, no global state -- like the reference's re-entrant extension, _pwm.c:7-70;
ctypes releases the GIL, so the calls really overlap).  Results must equal the oracle's whatever the interleaving."""
import threading

import numpy as np
import pytest

from conftest import assert_f32_bits_equal, assert_struct_close
from test_gpu_parity import rand_stream, rand_struct_pssm, rand_table

pytestmark = pytest.mark.gpu


def test_two_threads_two_contexts(oracle):
    from rnascan_amd import _lib
    jobs = []
    for t in range(2):
        rng = np.random.default_rng(800 + t)
        m = (8, 12)[t]
        s = rand_stream(rng, 60, 300, 3000)
        T, P = rand_table(rng, m), rand_struct_pssm(rng, m)
        want_seq, want_st = oracle.stream_seq(s.codes, T), oracle.stream_struct(s.profile, P)
        fin = want_seq[np.isfinite(want_seq)].astype(np.float64)
        thrs = [float(np.quantile(fin, q)) for q in (0.999, 0.99, 0.9, 0.9995)]
        jobs.append((s, T, P, want_seq, want_st, thrs))
    errors = []

    def work(t):
        try:
            s, T, P, want_seq, want_st, thrs = jobs[t]
            with _lib.Context(0) as c:
                both, only = c.motif(T, P), c.motif(letter_table=T)
                for it in range(12):
                    thr = thrs[it % len(thrs)]
                    pos, sq, _ = c.hits_host(only, s.codes, thr_seq=thr)          # credit prefilter: the per-motif threshold cache
                    want_pos = oracle.stream_hits(want_seq, None, thr, -np.inf)
                    assert np.array_equal(pos, want_pos), (t, it, pos.size, want_pos.size)
                    assert_f32_bits_equal(sq, want_seq[want_pos])
                    got_seq, got_st = c.scan_host(both, s.codes, s.profile)       # k_profile
                    assert_f32_bits_equal(got_seq, want_seq)
                    assert_struct_close(got_st, want_st)
                    pos, sq, st = c.hits_host(both, s.codes, s.profile, thr_seq=thr, thr_struct=-3.0)
                    want_pos = oracle.stream_hits(want_seq, want_st, thr, -3.0)
                    assert np.array_equal(pos, want_pos), (t, it, "combined")
                    assert_struct_close(st, want_st[want_pos])
                both.close()
                only.close()
        except BaseException as e:                                                # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in threads), "a scanning thread did not finish"
    assert not errors, errors
