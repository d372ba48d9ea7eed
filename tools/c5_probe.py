"""C5 probe: 256 seq+struct PFM pairs over the resident C3 stream through the one-pass library kernel."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from rnascan_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--records", type=int, default=100000)
ap.add_argument("--motifs", type=int, default=256)
ap.add_argument("--width", type=int, default=12)
ap.add_argument("--thr-seq", type=float, default=6.0)
ap.add_argument("--thr-struct", type=float, nargs="+", default=[-12.0])
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--seq-only", action="store_true")
ap.add_argument("--profile-dtype", choices=["float32", "float64"], default="float32")
args = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
codes, profile, n_pos = bench.make_stream(torch, dev, args.records, 3000, 20240601)
T, P = [], []
for k in range(args.motifs):
    t, p = bench.make_pssms(args.width, "finite", seed=1000 + k)
    T.append(t); P.append(p)
lib = ctx.library(np.stack(T), None if args.seq_only else np.stack(P))
ptype = _lib.PROFILE_F32
if args.profile_dtype == "float64":
    profile = profile.double()
    ptype = _lib.PROFILE_F64
cap = 1 << 25
hp = torch.empty(cap, dtype=torch.int64, device=dev); hm = torch.empty(cap, dtype=torch.int32, device=dev)
hs = torch.empty(cap, dtype=torch.float32, device=dev); ht = torch.empty(cap, dtype=torch.float64, device=dev)
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
win = args.records * (3000 - args.width + 1)
for thr_t in args.thr_struct:
    def step():
        ctx.library_hits_dev(lib, codes.data_ptr(), profile.data_ptr(), ptype, n_pos, args.thr_seq, thr_t, cap,
                             hp.data_ptr(), hm.data_ptr(), hs.data_ptr(), ht.data_ptr(), cnt.data_ptr(), st.cuda_stream)
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    print(json.dumps({"thr_seq": args.thr_seq, "thr_struct": thr_t, "ms": ms, "hits": int(cnt.item()), "hit_rate": int(cnt.item()) / (win * args.motifs),
                      "pairs_per_s": win * args.motifs / ms * 1e3, "motifs": args.motifs, "width": args.width, "profile_dtype": args.profile_dtype, "lib": os.path.basename(_lib.LIB_PATH), "info": lib.info()}))
