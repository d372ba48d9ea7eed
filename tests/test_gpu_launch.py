"""`--gpus N` from one command with the REAL engine, rehearsed on the one GPU of the box: two ranks share device 0
(gloo rendezvous: RCCL refuses two ranks on one device).  The N-GPU product launch differs only in the device each
rank takes and in the backend of bench.py's timing barrier."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA_DIR, REPO

pytestmark = pytest.mark.gpu
SEQ_PFM = os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt")


def _clean_env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


@pytest.mark.timeout(600)
def test_bench_gpus_2_rehearsal_reports_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--records", "4000", "--steps", "3",
                        "--warmup", "1", "--settle", "2"], env=_clean_env(PFMSCAN_BENCH_REHEARSE="1"),
                       capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and "dry_run" not in res
    assert len(res["per_rank"]["kernel_ms"]) == 2 and all(ms > 0 for ms in res["per_rank"]["kernel_ms"])
    assert res["per_rank"]["parity_sample_ok"] == [True, True]
    # what tells a slow rank from a badly placed one without a second run: every rank's placement line, spread of its launches,
    # the same kernel on default-allocator arrays and its own memory floor
    pr = res["per_rank"]
    for key in ("kernel_ms_median", "kernel_ms_min", "placement_note", "kernel_ms_default_allocator", "mixed_read_write_floor_ms"):
        assert len(pr[key]) == 2, key
    assert all(v > 0 for v in pr["kernel_ms_median"] + pr["kernel_ms_min"] + pr["kernel_ms_default_allocator"])
    assert all(isinstance(n, str) and n for n in pr["placement_note"])
    assert res["roofline"]["kernel_ms_default_allocator"] > 0
    # whole-job value: both ranks' windows over the slower rank's time
    assert res["value"] == pytest.approx(2 * 4000 * (3000 - 12 + 1) * 3 / (res["ms_per_step"] * 3e-3), rel=1e-6)


@pytest.mark.timeout(600)
def test_rnascan_gpus_2_equals_one_rank(tmp_path):
    rng = np.random.default_rng(9)
    fa = tmp_path / "many.fa"
    with open(fa, "w") as f:
        for i in range(300):
            f.write(">rec%d d%d\n%s\n" % (i, i, "".join(rng.choice(list("ACGTN"), size=int(rng.integers(0, 3000)),
                                                                     p=[.245, .245, .245, .245, .02]))))
    cmd = [sys.executable, os.path.join(REPO, "bin", "rnascan"), "-p", SEQ_PFM, "-C", "0.01", "-m", "2", str(fa)]
    one = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run(cmd + ["--gpus", "2"], env=_clean_env(RNASCAN_ONE_DEVICE="1"), capture_output=True, text=True, timeout=280)
    assert two.returncode == 0, two.stderr[-3000:]
    assert one.stdout.count("\n") > 50
    assert two.stdout == one.stdout


@pytest.mark.timeout(600)
@pytest.mark.parametrize("placement", ["tuned", "plain", "torch"])
def test_bench_placements_give_the_same_line_but_for_the_time(placement):
    """bench.py --placement tuned | plain | torch: the resident arrays come from pfmscan_place_alloc (measured / driver order) or
    from torch's allocator; the line says which, the parity sample is bit-exact either way and the value is that of its time"""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--records", "20000", "--steps", "3", "--warmup", "1", "--settle", "2",
                        "--no-secondary", "--no-ref-structured", "--cpu-seconds", "1", "--placement", placement],
                       env=_clean_env(PFMSCAN_BENCH_NO_FLOOR="1"), capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    pl = res["config"]["placement"]
    assert pl["mode"] == placement, pl
    if placement != "torch":
        assert ("NOT tuned" in pl["note"]) == (placement == "plain"), pl
    assert res["parity_on_sample"]["seq_f32_bit_exact"] is True and res["parity_on_sample"]["struct_within_1e-6"] is True
    assert res["roofline"]["kernel"] == "k_profile_fixed"
    assert res["value"] == pytest.approx(20000 * (3000 - 12 + 1) * 3 / (res["ms_per_step"] * 3e-3), rel=1e-6)
