"""Follow-up 2 of tools/placement_probe.py: all arrays of the headline scan inside ONE arena (one allocation), each moved
in turn by multiples of 8 MB .. 1 GB: which array's position matters, and with what period?   python tools/placement_arena.py [width]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from rnascan_amd import _lib
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    records, length = 100000, 3000
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    table, spssm = bench.make_pssms(width)
    motif = ctx.motif(table, spssm)
    stream = torch.cuda.Stream(device=dev)
    codes, profile0, n_pos = bench.make_stream(torch, dev, records, length, 1)
    GB = 1 << 30
    arena = torch.empty((int(sys.argv[2]) if len(sys.argv) > 2 else 64) * GB, dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    where = {"prof": None}

    def timed(po, so, to, steps=30, warm=12):
        if where["prof"] != po:
            arena[po:po + n_pos * 28].copy_(profile0.view(torch.uint8).view(-1))
            where["prof"] = po
        with torch.cuda.stream(stream):
            for _ in range(warm):
                ctx.scan_dev(motif, codes.data_ptr(), base + po, _lib.PROFILE_F32, n_pos, base + so, base + to, stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                ctx.scan_dev(motif, codes.data_ptr(), base + po, _lib.PROFILE_F32, n_pos, base + so, base + to, stream.cuda_stream)
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    A = arena.numel() // GB
    P, S, T = 0, 10 * GB, 12 * GB
    print("arena %d GB at %#x; base %.4f" % (A, base, timed(P, S, T)), flush=True)
    grid = list(range(16, A - 4, 4))
    print("out_seq at g GB (profile 0, out_struct 12):", " ".join("%d:%.3f" % (g, timed(P, g * GB, T)) for g in grid), flush=True)
    best_s = min(grid, key=lambda g: timed(P, g * GB, T)) * GB
    print("out_struct at g GB (profile 0, out_seq %d):" % (best_s // GB), " ".join("%d:%.3f" % (g, timed(P, best_s, g * GB)) for g in grid if abs(g * GB - best_s) >= 4 * GB), flush=True)
    print("profile at g GB (out_seq 10, out_struct 12):", " ".join("%d:%.3f" % (g, timed(g * GB, S, T)) for g in grid if g * GB + 9 * GB < A * GB), flush=True)
    print("base again %.4f" % timed(P, S, T), flush=True)


if __name__ == "__main__":
    main()
