import json
import os
import sys

import numpy as np
import pytest

# the oracle's OpenMP loops run on arrays of a few thousand positions here: with one thread per visible CPU (128+ on
# the GPU box, of which the job may use 16) every parallel region costs more in spin-waiting than the loop itself
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
DATA_DIR = os.path.join(GOLDEN_DIR, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def data_dir():
    return DATA_DIR


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session (fails loudly without the library)."""
    from rnascan_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def assert_f32_bits_equal(got, want):
    """bit-exact float32 comparison with NaN == NaN (payload/sign of NaN ignored)."""
    got = np.asarray(got, dtype=np.float32)
    want = np.asarray(want, dtype=np.float32)
    assert got.shape == want.shape
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), "NaN pattern differs at %s" % np.flatnonzero(gn != wn)[:10]
    ok = got.view(np.uint32)[~gn] == want.view(np.uint32)[~wn]
    assert ok.all(), "float32 bits differ at %s" % np.flatnonzero(~ok)[:10]


def assert_struct_close(got, want, tol=1e-6):
    """structure scores: |d| <= 1e-6 absolute (north star), exact class for NaN/inf,
    relative 1e-12 for the +-DBL_MAX-sized values nan_to_num produces."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN pattern differs"
    inf = np.isinf(want)
    assert np.array_equal(got[inf], want[inf]), "inf pattern differs"
    fin = np.isfinite(want)
    big = fin & (np.abs(want) > 1e9)
    small = fin & ~big
    if small.any():
        err = np.abs(got[small] - want[small])
        assert err.max() <= tol, "max abs err %.3e" % err.max()
    if big.any():
        assert np.allclose(got[big], want[big], rtol=1e-12, atol=0), "huge values differ"


def nasty_fasta(path, n=40, seed=3):
    """records with CRLF, wrapped and blank lines, lower case, T for U, foreign letters, blanks and tabs in the lines,
    headers that need csv quoting -- some of them holding the SLBP site so that there are hits to format"""
    rng = np.random.default_rng(seed)
    site = "AAAGGCTCTTTTCAGAGC"
    with open(path, "wb") as f:
        f.write(b"; a comment line before the first record\n")
        for i in range(n):
            body = "".join("ACGT"[c] for c in rng.integers(0, 4, size=int(rng.integers(0, 200))))
            if i % 3 == 0:
                body = body[:50] + site + body[50:] + (site.lower() if i % 2 else "")
            if i % 5 == 0:
                body = body[:20] + "N" + body[20:]
            eol = "\r\n" if i % 4 == 0 else "\n"
            header = ">rec%d description %d" % (i, i) + ('\twith a "tab"' if i % 7 == 0 else "") + ("  " if i % 6 == 0 else "")
            lines = [body[k:k + 60] for k in range(0, len(body), 60)] or [""]
            if i % 8 == 0:
                lines[0] = " " + lines[0][:10] + " " + lines[0][10:] + "\t"
            f.write((header + eol + eol.join(lines) + eol + (eol if i % 9 == 0 else "")).encode("ascii"))
