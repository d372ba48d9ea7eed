"""ref_structured.py -- TEST / BENCH INFRASTRUCTURE ONLY.

"B-ref" of BASELINE.md section 3: the reference's COST MODEL restated, to be timed next
to the GPU numbers.  It does what rnascan does per record, the way rnascan does it:

  * sequence side (rnascan.py:258-275 through Biopython's per-window ``search``): a
    Python loop over window starts, one ``calculate(window)`` call per window -- here one
    call into the C oracle per window (the reference calls its C extension per window,
    after rebuilding the m x 4 list-of-lists, matrix.py:57-60), strict ``>`` filter,
    ``round(score, 3)``, list append, one DataFrame per record, sorted;
  * averaged-structure side (rnascan.py:293-315): pandas ``iloc`` row slices,
    ``np.dot`` + ``np.nan_to_num`` per (window, row), one ``pd.Series`` per hit;
  * fan-out (rnascan.py:363-366, :388-395): ``multiprocessing.Pool(cores)`` mapped over
    records in batches of 2000.

Only bench.py's cpu_baseline leg and tests import this.
"""
import multiprocessing
import time
from itertools import repeat

import numpy as np
import pandas as pd

from . import oracle


def _scan_seq_record(args):
    seq, matrix, minscore = args
    m = matrix.shape[0]
    L = oracle.lib()
    out = np.empty(1, dtype=np.float32)
    out_p = out.ctypes.data
    rows = []
    logodds = [[float(matrix[i, c]) for c in range(4)] for i in range(m)]
    for pos in range(len(seq) - m + 1):
        window = seq[pos:pos + m]
        M = np.array(logodds)                      # the list-of-lists -> ndarray conversion of every call
        L.oracle_pwm_calculate(window.encode("ascii"), m, M.ctypes.data, m, out_p)   # one C call per window
        score = out[0]
        if score > minscore:
            rows.append(["motif", pos + 1, pos + m, window, round(score, 3)])
    df = pd.DataFrame(rows, columns=["Motif_ID", "Start", "End", "Sequence", "LogOdds"])
    return df.sort_values(["Start", "Motif_ID"])


def _scan_struct_record(args):
    profile, pssm, minscore = args
    struct = pd.DataFrame(profile)
    pm = pd.DataFrame(pssm)
    N = len(pm.index)
    hits = []
    for i in range(0, len(struct.index) - N + 1):
        score = 0
        for j in range(0, N):
            score += np.nan_to_num(np.dot(struct.iloc[i + j, :], pm.iloc[j, :]))
        if score > minscore:
            hits.append(pd.Series(["motif", i + 1, i + N, ".", score],
                                  index=["Motif_ID", "Start", "End", "Sequence", "LogOdds"]))
    return pd.DataFrame(hits)


def time_reference_structured(seqs, profiles, matrix, pssm, minscore, cores):
    """seconds to scan the given records the reference's way with a Pool of `cores`;
    returns (seconds_seq, seconds_struct, n_seq_hits, n_struct_hits)."""
    with multiprocessing.Pool(cores) as p:
        t0 = time.perf_counter()
        a = p.map(_scan_seq_record, zip(seqs, repeat(matrix), repeat(minscore)))
        t1 = time.perf_counter()
        b = p.map(_scan_struct_record, zip(profiles, repeat(pssm), repeat(minscore)))
        t2 = time.perf_counter()
    return t1 - t0, t2 - t1, sum(len(x) for x in a), sum(len(x) for x in b)
