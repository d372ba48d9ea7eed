"""One command, N GPUs: the parent that fans a run out over the GPUs of a node.

The reference's fan-out is one flag: ``-c N`` gives ``multiprocessing.Pool(N)`` workers that each take
records (rnascan/rnascan.py:66-67, :363-366, :388-395).  Here the unit is one process per GPU: the parent
-- which never touches a GPU (no HIP call, no ``torch.cuda``, no ``_lib.Context``) -- starts N fresh child
processes with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), waits
for them and returns rank 0's standard output.  A child that fails ends the others and makes the parent
return non-zero.  Children see WORLD_SIZE and therefore never fan out again.  Nothing is re-executed in
place: a process that has initialised the GPU is never replaced by another program.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    """the environment of one rank (what ``python -m torch.distributed.run --nnodes=1`` would set)"""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between processes on this driver
    return env


def spawn_ranks(world, cmd, capture_rank0=False, extra_env=None, poll=0.05, grace=10.0):
    """Start ``cmd`` (a list) once per rank and wait.  Rank 0 keeps the parent's stdout (``capture_rank0``: its
    stdout is collected and returned instead); the other ranks' stdout goes to the parent's stderr, so that
    stdout carries exactly what one rank would have printed.  Returns (exit code, rank 0's stdout or None):
    0 when every rank returned 0, else the first failing rank's code (the others are terminated)."""
    if world < 1:
        raise ValueError("world must be positive")
    port = free_port()
    procs, box, reader = [], [b""], None
    try:
        for rank in range(world):
            env = rank_env(rank, world, port)
            if extra_env:
                env.update(extra_env)
            if rank == 0:
                out = subprocess.PIPE if capture_rank0 else None
            else:
                try:
                    out = sys.stderr.fileno()
                except (AttributeError, OSError, ValueError):      # a captured / replaced stderr without a descriptor
                    out = subprocess.DEVNULL
            procs.append(subprocess.Popen(cmd, env=env, stdout=out))
        if capture_rank0:
            def pump():
                box[0] = procs[0].stdout.read()
            reader = threading.Thread(target=pump, daemon=True)
            reader.start()
        rc = 0
        live = set(range(world))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code      # a signal's negative code -> the shell's 128 + n
                    sys.stderr.write("rank %d of %d exited with code %d; stopping the other ranks\n" % (r, world, code))
                    break
            if live and rc == 0:
                time.sleep(poll)
        text = None
        if capture_rank0 and rc == 0:
            reader.join()
            text = box[0].decode("utf-8", "replace")
        return rc, text
    finally:
        # exactly the processes started here, by PID: first a polite signal, then the hard one
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + grace
        for p in procs:
            while p.poll() is None and time.time() < t_end:
                time.sleep(poll)
            if p.poll() is None:
                p.kill()
                p.wait()
        if reader is not None:
            reader.join(timeout=grace)


def resolve_world(gpus):
    """(world, must_spawn): ``gpus`` is the value of --gpus (None = not given).  Under a launcher (WORLD_SIZE set)
    the two must agree; without one, --gpus N > 1 asks THIS process to become the parent of N ranks."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        world = int(env_world)
        if gpus is not None and gpus != world:
            raise SystemExit("--gpus %d but this process runs as one of WORLD_SIZE=%d ranks: launch with "
                             "--nproc-per-node %d, or drop the outer launcher and let --gpus start the ranks" % (gpus, world, gpus))
        return world, False
    world = 1 if gpus is None else int(gpus)
    if world < 1:
        raise SystemExit("--gpus must be at least 1")
    return world, world > 1
