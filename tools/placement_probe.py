"""Does the headline kernel's time depend on WHERE its four arrays lie in HBM?  One process, one build, one width:
allocate the C3 arrays `trials` times (the earlier sets stay allocated, so every set gets other physical memory), time the
same launches on every set, then time the first set again (what moved with time rather than with placement).

    python tools/placement_probe.py [width] [trials] [records] [length]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from rnascan_amd import _lib
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    records = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
    length = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    table, spssm = bench.make_pssms(width)
    motif = ctx.motif(table, spssm)
    stream = torch.cuda.Stream(device=dev)
    codes0, profile0, n_pos = bench.make_stream(torch, dev, records, length, 1)
    sets = []

    def timed(s, steps=60, warm=25):
        codes, profile, out_seq, out_st = s
        with torch.cuda.stream(stream):
            for _ in range(warm):
                ctx.scan_dev(motif, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, out_seq.data_ptr(), out_st.data_ptr(), stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                ctx.scan_dev(motif, codes.data_ptr(), profile.data_ptr(), _lib.PROFILE_F32, n_pos, out_seq.data_ptr(), out_st.data_ptr(), stream.cuda_stream)
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    for t in range(trials):
        if t == 0:
            codes, profile = codes0, profile0
        else:
            codes, profile = codes0.clone(), profile0.clone()
        out_seq = torch.empty(n_pos, dtype=torch.float32, device=dev)
        out_st = torch.empty(n_pos, dtype=torch.float64, device=dev)
        s = (codes, profile, out_seq, out_st)
        sets.append(s)
        ms = timed(s)

        def op_ms(fn, n):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        rd = op_ms(lambda: profile.view(-1).sum(), 5)                 # each array on its own: read the profile, fill the outputs
        w8 = op_ms(lambda: out_st.zero_(), 10)
        w4 = op_ms(lambda: out_seq.zero_(), 10)
        print("       alone: profile.sum %.3f ms = %.2f TB/s   out_struct.zero_ %.3f ms = %.2f TB/s   out_seq.zero_ %.3f ms = %.2f TB/s"
              % (rd, profile.numel() * 4 / rd * 1e-9, w8, out_st.numel() * 8 / w8 * 1e-9, w4, out_seq.numel() * 4 / w4 * 1e-9))
        print("set %d  %.4f ms   codes %#x profile %#x out_seq %#x out_struct %#x" % (t, ms, codes.data_ptr(), profile.data_ptr(), out_seq.data_ptr(), out_st.data_ptr()), flush=True)
    for rnd in range(2):
        print("again:", " ".join("%.4f" % timed(s) for s in sets), flush=True)


if __name__ == "__main__":
    main()
