"""Record sharding over the GPUs of one node.

Records are independent, the PSSM is a <1 KB constant, so N GPUs = N replicas of
the same kernel on disjoint contiguous record ranges: one process per GPU
(``torch.distributed``; backend nccl = RCCL on the GPU box, gloo in CPU tests),
NO collective on the data path.  The only exchange is the gather of the hit
tables onto rank 0, which keeps file order because the ranges are contiguous and
the gather concatenates in rank order; ``Match_ID`` is numbered afterwards
(rnascan.py:329-332).  This replaces the reference's
``multiprocessing.Pool.map`` fan-out (rnascan.py:363-366, :388-395).
"""
import os

import numpy as np
import pandas as pd


def partition(lengths, world):
    """Contiguous record ranges [(lo, hi), ...] (one per rank) balanced by the
    number of stream positions sum(L_r + 1); ranges may be empty."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if world <= 0:
        raise ValueError("world must be positive")
    cost = np.cumsum(lengths + 1)
    total = int(cost[-1]) if n else 0
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        cut = int(np.searchsorted(cost, target, side="left"))
        if cut < n and (cut == 0 or abs(cost[cut] - target) < abs(cost[cut - 1] - target)):
            cut += 1
        bounds.append(max(bounds[-1], min(cut, n)))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def gather_frames(df, rank, world, dist=None):
    """Concatenate per-rank hit tables on rank 0 in rank order (= file order).
    Returns the full table on rank 0 and None elsewhere."""
    if world == 1:
        return df
    if dist is None:
        import torch.distributed as dist
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(df, bucket, dst=0)
    if rank != 0:
        return None
    frames = [f for f in bucket if f is not None and len(f)]
    if not frames:
        return bucket[0]
    return pd.concat(frames, ignore_index=True)


def batches(lengths, lo, hi, max_positions):
    """Cut records [lo, hi) into contiguous batches of at most ``max_positions`` stream
    positions sum(L_r + 1) each (a record longer than that is a batch of its own)."""
    cost = np.cumsum(np.asarray(lengths[lo:hi], dtype=np.int64) + 1)       # cost[i] = positions of records lo .. lo + i
    out, start, done = [], 0, 0                                            # start: first record of the open batch (relative)
    n = hi - lo
    while start < n:
        # the last record r with cost[r] - done <= max_positions; at least one record per batch
        end = int(np.searchsorted(cost, done + max_positions, side="right"))
        end = max(end, start + 1)
        out.append((lo + start, lo + end))
        done = int(cost[end - 1])
        start = end
    if not out:
        out.append((lo, hi))
    return out


def batch_positions():
    """Stream positions one launch may hold (RNASCAN_BATCH_POSITIONS, default 2**24 = 16.8 M:
    0.94 GB of float64 profile rows on the device and in the packed host copy, and at most that many
    table rows in memory at `-m ' -inf'`).  Larger inputs go through several launches (a launch + its
    read-back cost well under a millisecond); the tables are written / concatenated in record order."""
    return max(1, int(os.environ.get("RNASCAN_BATCH_POSITIONS", str(1 << 24))))


def scan_sharded(items, lengths, scan_fn, rank=None, world=None, dist=None, max_positions=None, sink=None):
    """Run ``scan_fn`` over this rank's contiguous range -- one call per batch of at
    most ``max_positions`` stream positions -- and gather the tables on rank 0.
    ``items`` is any sliceable list (records, profiles or pairs of them), ``lengths``
    their lengths.  With a ``sink`` every batch's table is handed to ``sink(frame)`` as soon
    as it exists and nothing is kept or gathered (returns None on every rank): the hit table
    never has to fit in memory -- a multi-rank caller gives every rank its own sink and joins the
    ranks' outputs itself (``relay_spools``)."""
    if rank is None or world is None:
        rank, world = env_rank_world()
    lo, hi = partition(lengths, world)[rank]
    parts = batches(lengths, lo, hi, max_positions or batch_positions())
    if sink is not None:
        for a, b in parts:
            sink(scan_fn(items[a:b]))
        return None
    frames = [scan_fn(items[a:b]) for a, b in parts]
    if len(frames) == 1:
        local = frames[0]
    else:
        full = [f for f in frames if f is not None and len(f)]
        local = pd.concat(full, ignore_index=True) if full else frames[0]
    return gather_frames(local, rank, world, dist)


# ---------------------------------------------------------------------------
# multi-rank output without a table in memory anywhere
# ---------------------------------------------------------------------------
SPOOL_BLOCK = 32 << 20


def open_spool():
    """a rank's row spool: a temporary file (RNASCAN_SPOOL_DIR, else the system's) that rank 0 reads back; text mode over
    a byte stream, like stdout, so that the TSV writer takes its zero-copy path"""
    import atexit
    import tempfile
    fd, path = tempfile.mkstemp(prefix="rnascan_rows_", suffix=".tsv", dir=os.environ.get("RNASCAN_SPOOL_DIR") or None)

    def drop(path=path):                       # a rank that dies before the relay (launch.py ends the others) leaves no spool behind
        try:
            os.unlink(path)
        except OSError:
            pass
    atexit.register(drop)
    return os.fdopen(fd, "w", encoding="utf-8", newline=""), path


def relay_spools(out, first_id, my_path, my_rows, rank, world, dist):
    """Every rank > 0 has written its rows -- formatted, WITHOUT the Match_ID column -- to its spool file.  Rank 0 appends
    them to ``out`` in rank order (= record order: the ranges are contiguous) and numbers them while it copies
    (pfmscan_tsv_number), so that ``Match_ID`` is 1..n over the whole table as rnascan.py:329-332 makes it -- and no rank
    ever holds more than one batch of rows.  The ranks of one node share a filesystem; a spool rank 0 cannot open is sent
    over the process group in blocks instead.  Returns the next free id on rank 0 (None elsewhere)."""
    import torch
    from . import _lib
    # the blocks of the send path are tensors of the process group's backend: NCCL (RCCL) moves device memory only
    nccl = getattr(dist, "get_backend", lambda: "gloo")() == "nccl"
    where = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    infos = [None] * world if rank == 0 else None
    dist.gather_object((my_path, int(my_rows)), infos, dst=0)
    readable = [None]
    if rank == 0:
        shared = os.environ.get("RNASCAN_SPOOL_SEND") != "1"       # "1": never read another rank's file (tests the send path)
        readable[0] = [r == 0 or infos[r][1] == 0 or (shared and os.access(infos[r][0], os.R_OK)) for r in range(world)]
    dist.broadcast_object_list(readable, src=0)
    readable = readable[0]
    if rank != 0:
        if my_rows and not readable[rank]:
            with open(my_path, "rb") as f:
                while True:
                    block = f.read(SPOOL_BLOCK)
                    dist.send(torch.tensor([len(block)], dtype=torch.int64, device=where), dst=0)
                    if not block:
                        break
                    dist.send(torch.frombuffer(bytearray(block), dtype=torch.uint8).to(where), dst=0)
        dist.barrier()                                     # rank 0 is done with the file
        if my_path and os.path.exists(my_path):
            os.unlink(my_path)
        return None
    raw = getattr(out, "buffer", None)
    if raw is not None:
        out.flush()
    next_id = int(first_id)
    import codecs
    text = codecs.getincrementaldecoder("utf-8")()         # a block may end inside a multi-byte character

    def emit(block, state):
        nonlocal next_id
        data, rows, state = _lib.tsv_number(block, next_id, state)
        next_id += rows
        if raw is not None:
            raw.write(data)
        else:
            out.write(text.decode(data))
        return state

    for r in range(1, world):
        path, rows = infos[r]
        if not rows:
            continue
        state = 0
        if readable[r]:
            with open(path, "rb") as f:
                while True:
                    block = f.read(SPOOL_BLOCK)
                    if not block:
                        break
                    state = emit(block, state)
        else:
            size = torch.zeros(1, dtype=torch.int64, device=where)
            while True:
                dist.recv(size, src=r)
                if int(size[0]) == 0:
                    break
                buf = torch.empty(int(size[0]), dtype=torch.uint8, device=where)
                dist.recv(buf, src=r)
                state = emit(buf.cpu().numpy().tobytes(), state)
    dist.barrier()
    return next_id
