"""How fast do profile rows get from a page-cache-warm mapped file onto the device?  usage: python tools/mmap_upload_probe.py [GB]
 (a) pfmscan_stage straight from the mapping, (b) after touching its pages with 16 threads, (c) from anonymous memory,
 (d) pread by 16 threads into anonymous memory, then stage."""
import os, sys, time, tempfile, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rnascan_amd import _lib

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n_pos = int(gb * 1e9 / 28)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
path = os.path.join(d, "p.f32")
rng = np.random.default_rng(0)
with open(path, "wb") as f:
    for lo in range(0, n_pos, 1 << 22):
        f.write(rng.random((min(1 << 22, n_pos - lo), 7), dtype=np.float32).tobytes())
ctx = _lib.Context(0)
anon = np.fromfile(path, dtype=np.float32).reshape(-1, 7)


def timed(label, fn):
    t = time.time()
    fn()
    dt = time.time() - t
    print("%-46s %.3f s  %.1f GB/s" % (label, dt, n_pos * 28 / dt / 1e9))


def touch(arr, threads=16):
    flat = arr.reshape(-1)
    step = 1024                                       # one float per 4-KiB page
    cuts = np.linspace(0, flat.size, threads + 1).astype(np.int64)
    ts = [threading.Thread(target=lambda a, b: flat[a:b:step].sum(), args=(cuts[i], cuts[i + 1])) for i in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]


def pread_into(out, threads=16):
    fd = os.open(path, os.O_RDONLY)
    raw = memoryview(out.reshape(-1).view(np.uint8))
    cuts = np.linspace(0, len(raw), threads + 1).astype(np.int64) // 4096 * 4096
    cuts[-1] = len(raw)

    def work(a, b):
        pos = a
        while pos < b:
            pos += os.preadv(fd, [raw[pos:min(b, pos + (64 << 20))]], pos)
    ts = [threading.Thread(target=work, args=(int(cuts[i]), int(cuts[i + 1]))) for i in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    os.close(fd)


timed("(c) stage from anonymous memory (warm-up)", lambda: ctx.stage(None, anon))
timed("(c) stage from anonymous memory", lambda: ctx.stage(None, anon))
mm = np.memmap(path, dtype=np.float32, mode="r", shape=(n_pos, 7))
timed("(a) stage from a fresh mapping", lambda: ctx.stage(None, mm))
timed("(a) stage from the same mapping again", lambda: ctx.stage(None, mm))
mm2 = np.memmap(path, dtype=np.float32, mode="r", shape=(n_pos, 7))
timed("(b) touch a fresh mapping's pages, 16 threads", lambda: touch(mm2))
timed("(b) then stage from it", lambda: ctx.stage(None, mm2))
buf = np.empty((n_pos, 7), dtype=np.float32)
timed("(d) pread into anonymous memory, 16 threads (cold)", lambda: pread_into(buf))
timed("(d) pread into the same buffer again", lambda: pread_into(buf))
timed("(d) then stage from it", lambda: ctx.stage(None, buf))
