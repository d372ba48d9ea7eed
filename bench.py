#!/usr/bin/env python3
"""bench.py -- scored windows/s of the seq+struct sliding-window scan (w=12) on MI355X.

Workload (BASELINE.json configs[2], "C3"): 100k synthetic RNA records x 3 kb,
uint8 letter codes + averaged-structure profile (7 x float32 per position),
sequence PFM and structure PFM of width 12, all-scores mode (one float32 sequence
score + one float64 structure score per window).  A "step" is ONE pass of the hot
path over the whole batch, inputs already resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: records are independent, so every rank scans its own
100k-record shard (weak scaling), no data-path collective; only the timing
barrier and a MAX over ranks use RCCL.

Prints ONE JSON line (rank 0).  ``roofline`` is computed from the ALGORITHMIC
bytes (29 B read per position + 12 B written per window)
and the average kernel duration measured with HIP events on the launch stream.
``cpu_baseline`` times the CPU oracle (oracle/pfm_oracle.c, OpenMP) on a bounded
sample of the same records on this box's host cores, and the same sample is
used as a parity check of the GPU scores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_pssms(width, variant="finite", seed=0):
    """Seeded PFMs -> log-odds operands: rows ~ Dirichlet(0.5), uniform background.
    variant "finite": pseudocount 0.01 (every log-odds finite; SURVEY 8d C3 headline);
    variant "inf": pseudocount 0 with 15 % of the cells zeroed (-inf log-odds, the
    nan_to_num path of rnascan.py:306 becomes first-order behaviour)."""
    from rnascan_amd import pssm, pack
    from collections import OrderedDict
    rs = np.random.RandomState(11 + seed)
    seq_counts = rs.dirichlet(np.full(4, 0.5), size=width)
    rs = np.random.RandomState(13 + seed)
    st_counts = rs.dirichlet(np.full(7, 0.5), size=width)
    pc = 0.01
    if variant == "inf":
        pc = 0.0
        st_counts[np.random.RandomState(17).rand(width, 7) < 0.15] = 0.0
        seq_counts[np.random.RandomState(19).rand(width, 4) < 0.05] = 0.0
    seq = OrderedDict((l, seq_counts[:, k]) for k, l in enumerate("ACGU"))
    st = OrderedDict((l, st_counts[:, k]) for k, l in enumerate(pack.STRUCT_COLUMNS))
    seq_p = pssm.PSSM("ACGU", pssm.log_odds(pssm.normalize(seq, pc), None))
    st_p = pssm.PSSM(pack.STRUCT_COLUMNS, pssm.log_odds(pssm.normalize(st, pc), None))
    return seq_p.letter_table("ACGU"), st_p.matrix(pack.STRUCT_COLUMNS)


def make_stream(torch, dev, records, length, seed, foreign=0.0, zero_snap=False):
    """Synthetic records generated ON DEVICE: letters iid uniform over ACGU, profile
    rows ~ Dirichlet(0.3) stored float32; every record followed by one separator."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    stride = length + 1
    n_pos = records * stride
    codes = torch.randint(0, 4, (records, stride), dtype=torch.uint8, device=dev, generator=g)
    if foreign > 0:                      # letters outside the alphabet (e.g. N): poison the windows covering them
        codes[torch.rand((records, stride), device=dev, generator=g) < foreign] = 7
    codes[:, length] = 7
    profile = torch.empty((n_pos, 7), dtype=torch.float32, device=dev)
    chunk = 1 << 24
    for lo in range(0, n_pos, chunk):
        hi = min(n_pos, lo + chunk)
        x = torch._standard_gamma(torch.full((hi - lo, 7), 0.3, dtype=torch.float32, device=dev), generator=g)
        x.clamp_(min=1e-30)
        x = x / x.sum(dim=1, keepdim=True)
        if zero_snap:                    # exact zeros like real averaged-structure files (50 % of the example's cells)
            x[x < 0.02] = 0.0
            x = x / x.sum(dim=1, keepdim=True)
        profile[lo:hi] = x
        del x
    profile.view(records, stride, 7)[:, length, :] = 0
    return codes.view(-1), profile, n_pos


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle", type=int, default=15,
                    help="untimed launches right after data generation, BEFORE the --warmup steps: the chip's clocks "
                         "take ~10 launches to settle after the generation kernels (per-dispatch times ramp 3.1 -> 2.3 ms)")
    ap.add_argument("--records", type=int, default=100000, help="records per GPU")
    ap.add_argument("--length", type=int, default=3000)
    ap.add_argument("--width", type=int, default=12)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work for the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-structured", action="store_true",
                    help="skip the reference-structured Python baseline (B-ref of BASELINE.md section 3)")
    ap.add_argument("--workload", choices=["c3", "c2", "c5"], default="c3",
                    help="c3: seq+struct w=12 (the headline metric); c2: sequence-only PFM (BASELINE configs[1], use --width 8); "
                         "c5: a library of --motifs seq+struct PFM pairs over the same resident records, hits mode "
                         "(BASELINE configs[4]); value is then window x motif pairs per second")
    ap.add_argument("--motifs", type=int, default=256, help="PFM pairs of --workload c5")
    ap.add_argument("--variant", choices=["finite", "inf"], default="finite",
                    help="finite: pseudocount 0.01 PSSMs (headline); inf: pseudocount 0 PSSMs with -inf cells, profile with "
                         "exact zeros and 0.1 %% foreign letters (exercises nan_to_num / NaN windows at full size)")
    ap.add_argument("--mode", choices=["scores", "hits", "hits2"], default="scores",
                    help="scores: all-scores (the headline, 41.1 B/window); hits: thresholded hits, one fused pass "
                         "(29.1 B/window + 20 B/hit); hits2: candidate-then-verify (letters pass, structure only at its hits)")
    ap.add_argument("--profile-dtype", choices=["float32", "float64"], default="float32",
                    help="device storage of the profile rows (float32 = the headline, 41.1 B/window; float64 = the strict "
                         "variant of SURVEY 8d, 69.2 B/window)")
    ap.add_argument("--minscore", type=float, default=6.0, help="threshold of --mode hits (seq > m and struct > m)")
    args = ap.parse_args()

    import torch
    from rnascan_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libpfmscan has no CPU fallback)")
    # rehearsal on a one-GPU box: PFMSCAN_BENCH_REHEARSE=1 puts every rank on device 0 and rendezvous over gloo
    # (RCCL refuses two ranks on one device); the product launch is one rank per GPU over RCCL
    rehearse = os.environ.get("PFMSCAN_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    ctx = _lib.Context(local_rank)
    table, spssm = make_pssms(args.width, args.variant)
    seq_only = args.workload == "c2"
    motif = ctx.motif(table, None if seq_only else spssm)
    library = []
    if args.workload == "c5":
        args.mode = "hits2"
        for k in range(args.motifs):                      # seeds 1000 + k
            t_k, s_k = make_pssms(args.width, args.variant, seed=1000 + k)
            library.append(ctx.motif(t_k, s_k))
    codes, profile, n_pos = make_stream(torch, dev, args.records, args.length, 20240601 + rank,
                                        foreign=0.001 if args.variant == "inf" else 0.0,
                                        zero_snap=args.variant == "inf")
    ptype = _lib.PROFILE_F32
    if args.profile_dtype == "float64":
        profile = profile.double()
        ptype = _lib.PROFILE_F64
    # zero-filled (touched) outputs: first-touch of fresh device pages would otherwise
    # land in the first kernel launches and skew the per-kernel average rocprof reports
    out_seq = torch.zeros(n_pos, dtype=torch.float32, device=dev)
    out_st = torch.zeros(n_pos, dtype=torch.float64, device=dev)
    windows = args.records * (args.length - args.width + 1)
    # a real (non-null) torch stream: the ABI reads stream NULL as "the ctx's own
    # stream", and the HIP events below must sit on the stream the kernel runs on
    torch.cuda.synchronize()                 # generation ran on the default stream: finish it first
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0

    cap = 1 << 24
    if args.mode in ("hits", "hits2"):
        hit_pos = torch.empty(cap, dtype=torch.int64, device=dev)
        hit_seq = torch.empty(cap, dtype=torch.float32, device=dev)
        hit_st = torch.empty(cap, dtype=torch.float64, device=dev)
        hit_count = torch.zeros(1, dtype=torch.int64, device=dev)

    def step():
        if library:
            for mo in library:
                hit_count.zero_()
                ctx.hits_adaptive_dev(mo, codes.data_ptr(), profile.data_ptr(), ptype, n_pos, args.minscore,
                                      args.minscore, cap, hit_pos.data_ptr(), hit_seq.data_ptr(), hit_st.data_ptr(),
                                      hit_count.data_ptr(), stream)
            return
        if args.mode in ("hits", "hits2"):
            hit_count.zero_()
            (ctx.hits_dev if args.mode == "hits" else ctx.hits_adaptive_dev)(motif, codes.data_ptr(), None if seq_only else profile.data_ptr(),
                         _lib.PROFILE_NONE if seq_only else ptype, n_pos, args.minscore,
                         args.minscore, cap, hit_pos.data_ptr(), hit_seq.data_ptr(), hit_st.data_ptr(),
                         hit_count.data_ptr(), stream)
        elif seq_only:
            ctx.scan_dev(motif, codes.data_ptr(), None, _lib.PROFILE_NONE, n_pos, out_seq.data_ptr(), None, stream)
        else:
            ctx.scan_dev(motif, codes.data_ptr(), profile.data_ptr(), ptype, n_pos,
                         out_seq.data_ptr(), out_st.data_ptr(), stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    # settle: untimed launches until the clocks have ramped -- at least --settle of them AND at least 40 ms of
    # them (15 launches of the 0.3 ms sequence-only kernel are over before the ramp is: C2 read 0.33 ms with
    # them and 0.29 ms in steady state)
    settled = 0
    if not library:
        t_s = time.perf_counter()
        while settled < args.settle or (time.perf_counter() - t_s < 0.040 and settled < 5000):
            step()
            settled += 1
            if settled >= args.settle and settled % 16 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = e0.elapsed_time(e1) / args.steps          # HIP events on the launch stream
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    result = None
    if rank == 0:
        total_windows = windows * world * args.steps * (len(library) if library else 1)
        in_b, out_b, hit_b = (1, 4, 12) if seq_only else (29 if ptype == _lib.PROFILE_F32 else 57, 12, 20)
        alg_bytes = args.records * args.length * in_b + windows * out_b      # per launch, per GPU
        n_hits = None
        if args.mode != "scores":
            n_hits = int(hit_count.item())
            alg_bytes = args.records * args.length * in_b + min(n_hits, cap) * hit_b
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile) and args.mode == "scores" and args.profile_dtype == "float32":
            try:
                for tj in json.load(open(tfile)).get("entries", []):
                    if (tj.get("workload") == args.workload and tj.get("records") == args.records
                            and tj.get("length") == args.length and tj.get("width") == args.width):
                        traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": ("scored window x motif pairs/sec (%d-PFM library, seq+struct, w=%d)" % (len(library), args.width)) if library
                      else "scored windows/sec (%s, w=%d)" % ("seq-only" if seq_only else "seq+struct", args.width),
            "value": total_windows / elapsed,
            "unit": "window-motif pairs/s" if library else "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("C5: library of %d seq+struct PFM pairs (width %d) x %d synthetic records x %d nt per GPU, "
                             "resident stream, thresholded hits per motif (candidate-then-verify)"
                             % (len(library), args.width, args.records, args.length)) if library else
                            ("C2: %d synthetic RNA records x %d nt per GPU, sequence PFM width %d, uint8 codes"
                             % (args.records, args.length, args.width)) if seq_only else
                            ("C3: %d synthetic RNA records x %d nt per GPU, seq PFM + averaged-structure PFM width %d, "
                             "uint8 codes + %s [n][7] profile, all-scores (f32 seq + f64 struct per window)"
                             % (args.records, args.length, args.width, args.profile_dtype)),
                "records_per_gpu": args.records, "record_length": args.length, "pfm_width": args.width,
                "variant": args.variant, "settle_launches": settled, "windows_per_gpu_per_step": windows, "mode": "all-scores" if args.mode == "scores" else "hits",
                "minscore": None if args.mode == "scores" else args.minscore, "hits_per_step": n_hits,
                "sharding": "records, no collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": ("k_letters_pre" if args.mode != "scores" else "k_letters") if seq_only else "k_profile", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
                "frac_of_measured_copy_peak_6290": achieved / 6290.0,
            },
        }

        if args.mode == "hits2":
            # candidate-then-verify reads 1 B per position plus m rows per candidate: the fused-pass byte
            # count does not describe it, so no roofline figure is given for this mode
            result["roofline"] = None
        if world == 1 and not args.no_cpu_baseline and args.mode == "scores" and not seq_only:
            from oracle import oracle
            oracle.build()
            stride = args.length + 1
            probe = min(args.records, 500)

            def cpu_run(nrec):
                c = codes[: nrec * stride].cpu().numpy()
                p = profile[: nrec * stride].cpu().numpy()
                t = time.perf_counter()
                s1 = oracle.stream_seq(c, table)
                s2 = oracle.stream_struct(p, spssm)
                return time.perf_counter() - t, s1, s2

            dt, _, _ = cpu_run(probe)
            nrec = int(max(probe, min(args.records, probe * args.cpu_seconds / max(dt, 1e-6))))
            dt, ref_seq, ref_st = cpu_run(nrec)
            nwin = nrec * (args.length - args.width + 1)
            result["cpu_baseline"] = {
                "value": nwin / dt, "unit": "windows/s", "cores": oracle.num_threads(), "kind": "port",
                "sample": "first %d of %d records (%d windows), oracle/pfm_oracle.c stream_seq + stream_struct_f32, "
                          "OpenMP over positions, %.2f s" % (nrec, args.records, nwin, dt),
                "host_cpus": os.cpu_count(),
            }
            got_seq = out_seq[: nrec * stride].cpu().numpy()
            got_st = out_st[: nrec * stride].cpu().numpy()
            nan_ok = bool(np.array_equal(np.isnan(got_seq), np.isnan(ref_seq)))
            v = ~np.isnan(ref_seq)
            seq_ok = nan_ok and bool(np.array_equal(got_seq[v].view(np.uint32), ref_seq[v].view(np.uint32)))
            vs = np.isfinite(ref_st) & (np.abs(ref_st) < 1e9)
            st_err = float(np.abs(got_st[vs] - ref_st[vs]).max())
            big = ~vs & ~np.isnan(ref_st)
            big_ok = bool(np.array_equal(np.isnan(got_st), np.isnan(ref_st)) and
                          np.allclose(got_st[big], ref_st[big], rtol=1e-12, atol=0, equal_nan=True))
            result["parity_on_sample"] = {"seq_f32_bit_exact": seq_ok, "struct_max_abs_err": st_err,
                                          "struct_within_1e-6": bool(st_err <= 1e-6 and big_ok)}
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
            if not args.no_ref_structured:
                # B-ref: the reference's own cost structure (per-window Python loop + C call,
                # pandas iloc + np.dot + nan_to_num, multiprocessing.Pool over records) on a
                # small sample of the same records, scaled linearly in records (independent)
                from oracle import ref_structured
                cores = min(os.cpu_count() or 1, 128)
                n_seq, n_st = cores * 4, cores
                lut = np.array(list("ACGU") + ["N"] * 4)
                seqs, profs = [], []
                for r in range(n_seq):
                    c = codes[r * stride: r * stride + args.length].cpu().numpy()
                    seqs.append("".join(lut[c]))
                for r in range(n_st):
                    profs.append(profile[r * stride: r * stride + args.length].cpu().numpy().astype(np.float64))
                t_seq, t_st, _, _ = ref_structured.time_reference_structured(seqs, profs, table[:, :4].copy(), spssm, 6.0, cores)
                per_rec = t_seq / n_seq + t_st / n_st          # wall seconds per record with `cores` workers
                wpr = args.length - args.width + 1
                result["cpu_baseline_reference_structured"] = {
                    "value": wpr / per_rec, "unit": "windows/s", "cores": cores, "kind": "port",
                    "sample": "reference cost model (oracle/ref_structured.py): Pool(%d); sequence side %d records in %.2f s, "
                              "averaged-structure side %d records in %.2f s, scaled linearly in records"
                              % (cores, n_seq, t_seq, n_st, t_st),
                }
                result["speedup_vs_reference_structured"] = result["value"] / (wpr / per_rec)
        print(json.dumps(result))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
