"""The headline scan on arrays from torch's allocator, from pfmscan_place_alloc in driver order (plain) and from
pfmscan_place_alloc with the measured placement, in ONE process, each allocation twice.
    python tools/placement_ab.py [width] [records] [length]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from rnascan_amd import _lib
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    records = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    length = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    table, spssm = bench.make_pssms(width)
    motif = ctx.motif(table, spssm)
    stream = torch.cuda.Stream(device=dev)
    codes0, profile0, n_pos = bench.make_stream(torch, dev, records, length, 1)
    torch.cuda.synchronize()          # the generator ran on torch's stream, the scans run on ours
    want_seq = torch.empty(n_pos, dtype=torch.float32, device=dev)
    want_st = torch.empty(n_pos, dtype=torch.float64, device=dev)
    ctx.scan_dev(motif, codes0.data_ptr(), profile0.data_ptr(), _lib.PROFILE_F32, n_pos, want_seq.data_ptr(), want_st.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()

    def timed(c, p, s, t, steps=60, warm=25):
        with torch.cuda.stream(stream):
            for _ in range(warm):
                ctx.scan_dev(motif, c, p, _lib.PROFILE_F32, n_pos, s, t, stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                ctx.scan_dev(motif, c, p, _lib.PROFILE_F32, n_pos, s, t, stream.cuda_stream)
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    out_seq = torch.empty(n_pos, dtype=torch.float32, device=dev)
    out_st = torch.empty(n_pos, dtype=torch.float64, device=dev)
    print("torch allocator:      %.4f ms" % timed(codes0.data_ptr(), profile0.data_ptr(), out_seq.data_ptr(), out_st.data_ptr()), flush=True)
    for trial in range(3):
        for plain in (True, False):
            t0 = time.perf_counter()
            arrs = ctx.place_alloc([n_pos * 28, n_pos * 8, n_pos * 4, n_pos], plain=plain)
            dt = time.perf_counter() - t0
            time.sleep(float(os.environ.get("AB_SLEEP", "0")))
            pr, st, sq, co = [torch.as_tensor(a, device=dev) for a in arrs]
            pr.copy_(profile0.view(torch.uint8).view(-1))
            co.copy_(codes0)
            torch.cuda.synchronize()
            ok0 = bool(torch.equal(pr, profile0.view(torch.uint8).view(-1))) and bool(torch.equal(co, codes0))
            ms = timed(co.data_ptr(), pr.data_ptr(), sq.data_ptr(), st.data_ptr())
            print("   inputs intact right after the copy: %s" % ok0)
            assert pr.data_ptr() == arrs[0].ptr and st.data_ptr() == arrs[1].ptr, (pr.data_ptr(), arrs[0].ptr)
            same = bool(torch.equal(sq.view(torch.int32), want_seq.view(torch.int32)) and torch.equal(st.view(torch.int64), want_st.view(torch.int64)))
            if not same:
                a, b = sq.view(torch.int32), want_seq.view(torch.int32)
                bad = (a != b).nonzero()
                print("   float32 scores that differ: %d of %d, first at %s; profile copy intact: %s; codes intact: %s" % (
                    bad.numel(), a.numel(), bad[:3].flatten().tolist(), bool(torch.equal(pr, profile0.view(torch.uint8).view(-1))), bool(torch.equal(co, codes0))))
                a, b = st.view(torch.int64), want_st.view(torch.int64)
                bad = (a != b).nonzero()
                print("   fp64 scores that differ: %d, first at %s" % (bad.numel(), bad[:3].flatten().tolist()))
            print("place_alloc %-6s  %.4f ms   (allocation %.2f s; same bits as on torch arrays: %s)  %s" % ("plain" if plain else "tuned", ms, dt, same, ctx.place_note()), flush=True)
            del pr, st, sq, co
            ctx.place_free(arrs[0])
    print("torch allocator again: %.4f ms" % timed(codes0.data_ptr(), profile0.data_ptr(), out_seq.data_ptr(), out_st.data_ptr()), flush=True)


if __name__ == "__main__":
    main()
