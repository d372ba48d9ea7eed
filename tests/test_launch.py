"""`--gpus N` from ONE command (rnascan_amd/launch.py): the GPU-untouched parent starts N ranks, relays rank 0's
output and fails when a rank fails.  The hot path cannot run here (no GPU, and the product has no CPU path), so
bench.py runs its PFMSCAN_BENCH_DRYRUN plumbing and the CLI ranks run the oracle-backed TEST engine."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA_DIR, REPO


def _bench(args, env_extra, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_bench_gpus_2_spawns_two_ranks_and_relays_rank0():
    r = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], {"PFMSCAN_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line on stdout, rank 0's
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["dry_run"] is True and res["value"] is None
    assert len(res["per_rank"]["kernel_ms"]) == 2 and all(ms > 0 for ms in res["per_rank"]["kernel_ms"])


@pytest.mark.timeout(300)
def test_bench_failing_rank_fails_the_parent():
    r = _bench(["--gpus", "3", "--steps", "2"], {"PFMSCAN_BENCH_DRYRUN": "1", "PFMSCAN_BENCH_DRYRUN_FAIL_RANK": "2"})
    assert r.returncode == 3
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank 2 of 3 exited with code 3" in r.stderr


@pytest.mark.timeout(600)
def test_bench_gpus_8_dry_run_is_config_4s_sharding():
    """the shape the driver's 8-GPU run takes (no 8-GPU node can be had here): eight gloo ranks from one command, each
    with C4's 125 000 records x 3 kb (1M records over the node, rnascan.py:388-395 fans the same records out to -c
    workers), one JSON line with eight per-rank entries, the windows of ALL ranks in the value's numerator"""
    r = _bench(["--gpus", "8", "--steps", "3", "--warmup", "1"], {"PFMSCAN_BENCH_DRYRUN": "1"}, timeout=500)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["scaling"] == "weak" and len(res["per_rank"]["kernel_ms"]) == 8
    cfg = res["config"]
    assert cfg["records_per_gpu"] == 125000 and cfg["records_all_ranks"] == 1000000 and "C4: 1M records" in cfg["workload"]
    assert cfg["windows_per_gpu_per_step"] == 125000 * 2989 and cfg["windows_all_ranks_per_step"] == 8 * 125000 * 2989


@pytest.mark.timeout(600)
def test_bench_gpus_8_workload_c5_dry_run_and_a_failing_rank_stops_seven():
    """config 5's "8 x MI355X" form goes through the same launcher (`bench.py --gpus 8 --workload c5`: every rank scans the
    256-PFM library over its own records); and when one of eight ranks fails the parent stops the other seven, prints no
    result line and returns that rank's code"""
    r = _bench(["--gpus", "8", "--workload", "c5", "--steps", "2", "--warmup", "1"], {"PFMSCAN_BENCH_DRYRUN": "1"}, timeout=500)
    assert r.returncode == 0, r.stderr
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 8 and res["config"]["bench_workload"] == "c5"
    assert res["config"]["windows_all_ranks_per_step"] == 8 * 125000 * 2989 * 256           # window-motif pairs of all ranks
    r = _bench(["--gpus", "8", "--steps", "2"], {"PFMSCAN_BENCH_DRYRUN": "1", "PFMSCAN_BENCH_DRYRUN_FAIL_RANK": "5"}, timeout=500)
    assert r.returncode == 3
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank 5 of 8 exited with code 3" in r.stderr


def test_bench_gpus_must_match_an_outer_launcher():
    r = _bench(["--gpus", "8"], {"PFMSCAN_BENCH_DRYRUN": "1", "WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


@pytest.mark.timeout(300)
def test_bench_under_torchrun_as_the_driver_launches_it():
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2: the ranks see WORLD_SIZE and do not fan out"""
    from rnascan_amd import launch
    env = dict(os.environ, PFMSCAN_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(launch.free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def _write_fasta(path, n=23, seed=7):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for i in range(n):
            f.write(">rec%d desc %d\n%s\n" % (i, i, "".join(rng.choice(list("ACGT"), size=int(rng.integers(30, 500))))))


@pytest.mark.timeout(300)
def test_cli_gpus_2_prints_the_single_rank_table(tmp_path, monkeypatch, capfd):
    """rnascan --gpus 2 ...: the parent starts two ranks (gloo), rank 0's table reaches the parent's stdout unchanged"""
    import io
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from engines import OracleEngine
    from rnascan_amd import cli
    fa = tmp_path / "many.fa"
    _write_fasta(fa)
    argv = ["-p", os.path.join(DATA_DIR, "SLBP_pfm_assembled_normalized_seq.txt"), "-C", "0.01", "-m", "-1", str(fa)]
    single = io.StringIO()
    cli.main(argv, engine=OracleEngine(), out=single)
    assert single.getvalue().count("\n") > 10
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(cli, "RANK_COMMAND", [sys.executable, os.path.join(REPO, "tests", "cli_rank_main.py")])
    capfd.readouterr()
    rc = cli.main(argv + ["--gpus", "2"])
    got = capfd.readouterr()
    assert rc == 0, got.err
    assert got.out == single.getvalue()
    # a failing rank: the parent's code is non-zero and nothing is left running
    monkeypatch.setenv("RNASCAN_TEST_FAIL_RANK", "1")
    assert cli.main(argv + ["--gpus", "2"]) == 5
    capfd.readouterr()


def test_product_rank_command_is_the_package_itself():
    from rnascan_amd import cli
    assert cli.RANK_COMMAND[1:] == ["-m", "rnascan_amd"]
