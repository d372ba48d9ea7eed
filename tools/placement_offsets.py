"""Follow-up of tools/placement_probe.py: WITHIN one allocation, does shifting one array by a few KB..MB change the time of the
headline kernel?  (If the time depends on the arrays' relative physical alignment, a shift inside a physically contiguous
allocation moves that alignment.)   python tools/placement_offsets.py [width]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch
    import bench
    from rnascan_amd import _lib
    width = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    records, length = 100000, 3000
    dev = torch.device("cuda:0")
    ctx = _lib.Context(0)
    table, spssm = bench.make_pssms(width)
    motif = ctx.motif(table, spssm)
    stream = torch.cuda.Stream(device=dev)
    codes, profile0, n_pos = bench.make_stream(torch, dev, records, length, 1)
    PAD = 8 << 20
    prof_buf = torch.empty(n_pos * 28 + PAD, dtype=torch.uint8, device=dev)
    seq_buf = torch.empty(n_pos * 4 + PAD, dtype=torch.uint8, device=dev)
    st_buf = torch.empty(n_pos * 8 + PAD, dtype=torch.uint8, device=dev)
    cur_prof_off = [None]

    def timed(po, so, to, steps=40, warm=15):
        if cur_prof_off[0] != po:
            prof_buf[po:po + n_pos * 28].copy_(profile0.view(torch.uint8).view(-1))
            cur_prof_off[0] = po
        p, s, t = prof_buf.data_ptr() + po, seq_buf.data_ptr() + so, st_buf.data_ptr() + to
        with torch.cuda.stream(stream):
            for _ in range(warm):
                ctx.scan_dev(motif, codes.data_ptr(), p, _lib.PROFILE_F32, n_pos, s, t, stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                ctx.scan_dev(motif, codes.data_ptr(), p, _lib.PROFILE_F32, n_pos, s, t, stream.cuda_stream)
            e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps

    offs = [0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 4 << 20]
    print("base %.4f" % timed(0, 0, 0), flush=True)
    print("out_struct shifted:", " ".join("%d:%.3f" % (o, timed(0, 0, o)) for o in offs), flush=True)
    print("out_seq shifted:   ", " ".join("%d:%.3f" % (o, timed(0, o, 0)) for o in offs), flush=True)
    print("profile shifted:   ", " ".join("%d:%.3f" % (o, timed(o, 0, 0)) for o in offs), flush=True)
    print("base again %.4f" % timed(0, 0, 0), flush=True)


if __name__ == "__main__":
    main()
