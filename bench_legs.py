"""bench_legs.py -- what bench.py's main() calls beside its timed region: the synthetic generators, the threshold probes, the
roofline helpers and the SECONDARY legs (the other BASELINE.json configs under the headline's clock, on the same resident
stream, after the headline's timed region).  bench.py keeps the contract: argument parsing, the resident stream, the timed
region of the headline kernel and the one JSON line; nothing here runs inside that region."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
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


def probe_motifs(tabs):
    """up to 8 motifs spread over a library (their score distributions differ)"""
    return [tabs[k] for k in range(0, len(tabs), max(1, len(tabs) // 8))][:8]


def struct_only_threshold(torch, ctx, tabs, profile, ptype, n_pos, out_st, stream, hit_rate):
    """structure-only library: the (1 - hit rate) quantile of the structure scores, mean over up to 8 motifs"""
    probe = probe_motifs(tabs)
    qs = []
    for tp in probe:
        m0 = ctx.motif(None, tp[1])
        ctx.scan_dev(m0, None, profile.data_ptr(), ptype, n_pos, None, out_st.data_ptr(), stream)
        torch.cuda.synchronize()
        sel = out_st[torch.isfinite(out_st)]
        k_top = max(2, int(round(hit_rate * sel.numel())))
        top = torch.topk(sel, k_top).values              # sorted descending
        qs.append(0.5 * (float(top[-1]) + float(top[-2])))       # between two scores, never ON one
        m0.close()
        del sel, top
    out_st.zero_()
    return float(np.mean(qs)), "auto: mean over %d motifs of the (1 - %g) quantile of their structure scores" % (len(probe), hit_rate)


def combined_threshold(torch, ctx, motifs, codes, profile, ptype, n_pos, out_seq, out_st, stream, windows, thr_seq, hit_rate):
    """the structure threshold that makes the COMBINED hit rate ``hit_rate`` at sequence threshold ``thr_seq`` (SURVEY 8d:
    "threshold chosen for ~1e-4"): a quantile of the structure scores of the windows passing the sequence threshold, pooled
    over ``motifs`` (open _lib.Motif objects).  Returns (threshold, fraction of windows passing the sequence side, note)."""
    pooled = []
    for m0 in motifs:
        ctx.scan_dev(m0, codes.data_ptr(), profile.data_ptr(), ptype, n_pos, out_seq.data_ptr(), out_st.data_ptr(), stream)
        torch.cuda.synchronize()
        sel = out_st[out_seq.double() > thr_seq]
        pooled.append(sel[torch.isfinite(sel)].clone())
    sel = torch.cat(pooled)
    rate_seq = float(sel.numel()) / (windows * len(motifs))
    keep = min(1.0, hit_rate / max(rate_seq, 1e-30))
    if sel.numel() == 0:
        thr = -1e30
    else:
        srt, _ = torch.sort(sel)
        i = min(sel.numel() - 1, int((1.0 - keep) * sel.numel()))
        # between two scores, never ON one: the kernels' structure sums differ in the last bit (FMA chain vs per-row
        # sum), and a threshold equal to a score would let that bit decide a hit
        thr = float(srt[i]) if i == 0 else 0.5 * (float(srt[i - 1]) + float(srt[i]))
    note = ("auto: quantile of the structure scores of the %.3g of windows with seq > %g (pooled over %d motif%s), for a "
            "combined rate of %g" % (rate_seq, thr_seq, len(motifs), "s" if len(motifs) > 1 else "", hit_rate))
    del sel, pooled
    out_seq.zero_()
    out_st.zero_()
    return thr, rate_seq, note


def pmc_entry(workload):
    """counter figures of an earlier rocprofv3 --pmc run of this workload (profiles/pmc_traffic.json), or {}"""
    try:
        for tj in json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json"))).get("entries", []):
            if tj.get("workload") == workload:
                return tj
    except Exception:
        pass
    return {}


def live_mixed_floor(records, length, placed=False, device=0, c2=False):
    """tools/hbm_mixed `quick` as a child process on the same GPU, right after the timed region: the fastest of its
    read+write forms is what THIS box's memory system gives the headline's byte mix (29 B in + 12 B out per position,
    nothing scored).  None when the tool is not built (rnascan_amd/build.py: build_floor_tool)."""
    import subprocess
    exe = os.path.join(REPO, "tools", "hbm_mixed")
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)
    if not os.access(exe, os.X_OK) or os.environ.get("PFMSCAN_BENCH_NO_FLOOR") or profiled:     # no child processes under a profiler
        return None
    try:
        env = dict(os.environ)
        vis = [v for v in env.get("HIP_VISIBLE_DEVICES", "").split(",") if v != ""]
        env["HIP_VISIBLE_DEVICES"] = vis[device] if device < len(vis) else str(device)          # the child sees only this rank's GPU
        out = subprocess.run([exe, str(records), str(length), "quick"] + (["placed"] if placed else []) + (["c2"] if c2 else []), capture_output=True, text=True,
                             timeout=120, env=env)
    except (OSError, subprocess.TimeoutExpired):
        return None
    for ln in out.stdout.splitlines():
        if ln.startswith("floor_ms "):
            f = ln.split()
            return {"ms": float(f[1]), "tb_per_s": float(f[3]), "bytes": float(f[5]),
                    "source": "tools/hbm_mixed %d %d quick%s%s, run by this bench.py on the same GPU after the timed region" % (
                        records, length, " placed" if placed else "", " c2" if c2 else ""),
                    "arrays": "pfmscan_place_alloc, like the bench's own" if placed else "hipMalloc, wherever the driver puts them",
                    "note": "fastest of 4 read+write forms (vector-load tiles, LDS-DMA tiles) in 3 rounds, each the median of 5 x 20 passes; "
                            "the figure moves by up to 10 % from one minute to the next on one box, like the kernel's own time"}
    return None


def library_roofline(info, windows, n_motifs, width, records, length, kernel_ms, n_hits, rate_seq):
    """k_library is bound by LDS look-ups, not by HBM: every window x motif group needs ceil(w/2) 16-byte table entries
    (ds_read_b128, 256 B/clk/CU).  ``achieved`` = the bytes phase A's look-ups MOVE per second (entries of 12 or 8
    motifs, padding motifs of the last group included); the nominal 2 B per motif credit is reported beside it, and so
    is what the counters of an earlier PMC run say about the LDS (the exact pass's gathers come on top of phase A)."""
    npair = (width + 1) // 2
    mpg = 12 if width <= 16 else 8
    groups = sum(-(-min(info["motifs_per_pass"], n_motifs - i * info["motifs_per_pass"]) // mpg) for i in range(info["passes"]))
    lds_read = float(windows) * groups * npair * 16
    nominal = float(windows) * n_motifs * npair * 2
    lds_peak = 256 * 256 * 2.4                      # CUs x B/clk/CU x GHz = GB/s (MI355X_MICROARCH.md, LDS)
    cand = windows * n_motifs * rate_seq if rate_seq is not None else None
    pmc = pmc_entry("c5")
    return {
        "bound": "lds", "achieved": lds_read / (kernel_ms * 1e-3) / 1e9, "peak": lds_peak, "unit": "GB/s",
        "frac": lds_read / (kernel_ms * 1e-3) / 1e9 / lds_peak, "traffic": lds_read,
        "traffic_source": "computed, not measured: 16-byte table entries of %d motifs read by phase A (LDS, not HBM, bytes); "
                          "the exact pass's gathers come on top" % mpg,
        "kernel": "k_library (%d passes of <= %d motifs; on a long stream they run side by side as teams of one launch)" % (
            info["passes"], info["motifs_per_pass"]),
        "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": lds_read,
        "algorithmic_unit": "LDS bytes the phase-A look-ups move: windows x motif groups x ceil(w/2) x 16 B",
        "nominal_2B_per_credit_frac": nominal / (kernel_ms * 1e-3) / 1e9 / lds_peak,
        "lds_idx_active_frac": pmc.get("lds_idx_active_frac"), "lds_bank_conflict_frac_of_active": pmc.get("lds_bank_conflict_frac"),
        "counter_source": pmc.get("source"),
        # HBM: from the counters of that PMC run (FETCH_SIZE + WRITE_SIZE per step, 1024-byte units; the gathers are 16-byte
        # loads, for which the gfx950 half-count correction of coalesced streams does not apply) over THIS run's time
        "hbm_frac": None if not pmc.get("hbm_bytes_per_step") else pmc["hbm_bytes_per_step"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "hbm_frac_source": pmc.get("source"),
        # what the kernel REQUESTS (codes once per pass + 28 B x w per candidate + the hits): mostly cache hits, NOT HBM traffic
        "requested_bytes_frac_of_hbm_peak": (records * length * info["passes"] + (cand or 0) * width * 28 + (n_hits or 0) * 24)
                                            / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "fp64_tflops_of_73_measured": None if cand is None else cand * width * 14 / (kernel_ms * 1e-3) / 1e12,
        "prefilter_slack_score_units": info["max_prefilter_eps"],
    }


def timed(torch, fn, steps, warmup):
    """ms per call of fn over ``steps`` calls after ``warmup`` untimed ones, HIP events on the current torch stream"""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def secondary_legs(torch, _lib, ctx, args, dev, codes, profile, ptype, n_pos, out_seq, out_st, stream, windows):
    """The other configs of BASELINE.json under the SAME clock as the headline, on the same resident stream, after the
    headline's timed region: C5 (256 seq+struct PFM pairs, k_library), its structure-only form (k_profile_lib) and C2
    in hits mode (sequence PFM of width 8 at the CLI's default -m 6, k_letters_cred).  A few steps each, ~2 s in all."""
    out = {}
    cap = 1 << 25
    hp = torch.empty(cap, dtype=torch.int64, device=dev)
    hm = torch.empty(cap, dtype=torch.int32, device=dev)
    hs = torch.empty(cap, dtype=torch.float32, device=dev)
    ht = torch.empty(cap, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    n_lib = 256
    tabs = [make_pssms(args.width, args.variant, seed=1000 + k) for k in range(n_lib)]
    # ---- C5
    motifs = [ctx.motif(*tp) for tp in probe_motifs(tabs)]
    thr_t, rate_seq, note = combined_threshold(torch, ctx, motifs, codes, profile, ptype, n_pos, out_seq, out_st, stream, windows, 6.0, 1e-4)
    for m0 in motifs:
        m0.close()
    lib = ctx.library(np.stack([t for t, _ in tabs]), np.stack([p for _, p in tabs]))

    def c5():
        ctx.library_hits_dev(lib, codes.data_ptr(), profile.data_ptr(), ptype, n_pos, 6.0, thr_t, cap, hp.data_ptr(), hm.data_ptr(),
                             hs.data_ptr(), ht.data_ptr(), cnt.data_ptr(), stream)
    ms = timed(torch, c5, 5, 2)
    hits = int(cnt.item())
    roof = library_roofline(lib.info(), windows, n_lib, args.width, args.records, args.length, ms, hits, rate_seq)
    out["c5"] = {"workload": "C5: %d seq+struct PFM pairs (w=%d) x the resident stream, k_library" % (n_lib, args.width),
                 "ms_per_step": ms, "steps": 5, "value": windows * n_lib / (ms * 1e-3), "unit": "window-motif pairs/s",
                 "hits_per_step": hits, "minscore_seq": 6.0, "minscore_struct": thr_t, "minscore_struct_source": note,
                 "lds_frac": roof["frac"], "roofline": roof}
    lib.close()
    # ---- C5, structure side alone: every motif in one pass over the profile
    thr_s, note_s = struct_only_threshold(torch, ctx, tabs, profile, ptype, n_pos, out_st, stream, 1e-4)
    slib = ctx.library(None, np.stack([p for _, p in tabs]))

    def c5s():
        ctx.library_hits_dev(slib, None, profile.data_ptr(), ptype, n_pos, None, thr_s, cap, hp.data_ptr(), hm.data_ptr(), None,
                             ht.data_ptr(), cnt.data_ptr(), stream)
    ms = timed(torch, c5s, 2, 1)
    tf = float(windows) * n_lib * args.width * 14 / (ms * 1e-3) / 1e12
    out["c5_struct_only"] = {"workload": "%d structure PFMs (w=%d) x the resident profile in ONE pass, k_profile_lib" % (n_lib, args.width),
                             "ms_per_step": ms, "steps": 2, "value": windows * n_lib / (ms * 1e-3), "unit": "window-motif pairs/s",
                             "hits_per_step": int(cnt.item()), "minscore_struct": thr_s, "minscore_struct_source": note_s,
                             "fp64_tflops": tf, "fp64_frac_of_73_measured": tf / 73.0}
    slib.close()
    # ---- C2 hits: sequence PFM of width 8, the CLI's default -m 6
    t8, _ = make_pssms(8, args.variant)
    m8 = ctx.motif(t8, None)
    w8 = args.records * (args.length - 8 + 1)

    def c2():
        cnt.zero_()
        ctx.hits_dev(m8, codes.data_ptr(), None, _lib.PROFILE_NONE, n_pos, 6.0, -np.inf, cap, hp.data_ptr(), hs.data_ptr(), None,
                     cnt.data_ptr(), stream)
    ms = timed(torch, c2, 20, 3)
    hits = int(cnt.item())
    gbs = (args.records * args.length + hits * 12) / (ms * 1e-3) / 1e9
    out["c2_hits"] = {"workload": "C2 hits: sequence PFM w=8 at -m 6 over the resident codes, k_letters_cred", "ms_per_step": ms, "steps": 20,
                      "value": w8 / (ms * 1e-3), "unit": "windows/s", "hits_per_step": hits, "minscore_seq": 6.0,
                      "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
    m8.close()
    # ---- SS hits: a structure letter-string PFM (7 letters, w = 12) at -m 6 over 100k x 3 kb structure strings, and the
    # two-FASTA combined scan (sequence PFM AND structure-letter PFM, both at -m 6) over the two code streams
    g = torch.Generator(device=dev)
    g.manual_seed(20240601 + 77)
    scodes = torch.randint(0, 7, (args.records, args.length + 1), dtype=torch.uint8, device=dev, generator=g)
    scodes[:, args.length] = 7
    scodes = scodes.view(-1)
    t12, p12 = make_pssms(12, args.variant)
    lt = np.full((12, 8), np.nan)
    lt[:, :7] = p12
    ms7, mq = ctx.motif(lt, None), ctx.motif(t12, None)
    w12 = args.records * (args.length - 12 + 1)

    def ss():
        cnt.zero_()
        ctx.hits_letters_f64_dev(ms7, scodes.data_ptr(), n_pos, 6.0, cap, hp.data_ptr(), ht.data_ptr(), cnt.data_ptr(), stream)
    ms = timed(torch, ss, 20, 3)
    hits = int(cnt.item())
    gbs = (args.records * args.length + hits * 16) / (ms * 1e-3) / 1e9
    out["ss_hits"] = {"workload": "SS hits: structure letter-string PFM (7 letters, w=12) at -m 6 over %d x %d structure strings, "
                                  "fp64 compare and score, k_letters_cred8" % (args.records, args.length),
                      "ms_per_step": ms, "steps": 20, "value": w12 / (ms * 1e-3), "unit": "windows/s", "hits_per_step": hits,
                      "minscore": 6.0, "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                      "vs_c2_hits": ms / out["c2_hits"]["ms_per_step"]}

    def pair():
        cnt.zero_()
        ctx.hits_pair_dev(mq, ms7, codes.data_ptr(), scodes.data_ptr(), n_pos, 6.0, 6.0, cap, hp.data_ptr(), hs.data_ptr(),
                          ht.data_ptr(), cnt.data_ptr(), stream)
    ms = timed(torch, pair, 10, 2)
    out["rnass_two_fasta"] = {"workload": "two-FASTA RNASS: sequence PFM (w=12) AND structure letter-string PFM (w=12) at -m 6 over the "
                                          "two code streams in ONE launch: k_letters_cred<.., PAIR> verifies the structure letters of its "
                                          "own survivors",
                              "ms_per_step": ms, "steps": 10, "value": w12 / (ms * 1e-3), "unit": "windows/s",
                              "hits_per_step": int(cnt.item()), "minscore": 6.0}
    ms7.close()
    mq.close()
    # ---- LETTER libraries (SURVEY 8f N1 x N4): 256 structure letter-string PFMs in ONE pass over the structure strings
    # (k_library8), and 256 (sequence PFM, structure-letter PFM) pairs over the two code streams (k_library, the structure
    # letters of its survivors) -- the reference would run its pure-Python _py_calculate once per motif (matrix.py:25-43)
    ST = np.full((n_lib, args.width, 8), np.nan)
    ST[:, :, :7] = np.stack([p for _, p in tabs])
    llib = ctx.library(None, struct_letters=ST)

    def ss_lib():
        ctx.library_hits_letters_dev(llib, scodes.data_ptr(), None, n_pos, None, 6.0, cap, hp.data_ptr(), hm.data_ptr(), None,
                                     ht.data_ptr(), cnt.data_ptr(), stream)
    ms = timed(torch, ss_lib, 3, 1)
    info = llib.info()
    rows = (args.width + 3) // 4 * 4
    groups = sum(-(-min(info["motifs_per_pass"], n_lib - i * info["motifs_per_pass"]) // 8) for i in range(info["passes"]))
    lds_read = float(windows) * groups * rows * 16
    out["ss_library"] = {"workload": "%d structure letter-string PFMs (7 letters, w=%d) at -m 6 x %d x %d structure strings in %d passes, "
                                     "k_library8 (single-letter 16-bit credits, 8 motifs per 16-byte entry; fp64 compare and score)"
                                     % (n_lib, args.width, args.records, args.length, info["passes"]),
                         "ms_per_step": ms, "steps": 3, "value": windows * n_lib / (ms * 1e-3), "unit": "window-motif pairs/s",
                         "hits_per_step": int(cnt.item()), "minscore": 6.0, "vs_c5": ms / out["c5"]["ms_per_step"],
                         "vs_per_motif_launches": ms / (n_lib * out["ss_hits"]["ms_per_step"]),
                         "lds_gbs": lds_read / (ms * 1e-3) / 1e9, "lds_frac": lds_read / (ms * 1e-3) / 1e9 / (256 * 256 * 2.4),
                         "prefilter_slack_score_units": info["max_prefilter_eps"]}
    llib.close()
    plib = ctx.library(np.stack([t for t, _ in tabs]), struct_letters=ST)

    def pair_lib():
        ctx.library_hits_letters_dev(plib, codes.data_ptr(), scodes.data_ptr(), n_pos, 6.0, 6.0, cap, hp.data_ptr(), hm.data_ptr(),
                                     hs.data_ptr(), ht.data_ptr(), cnt.data_ptr(), stream)
    ms = timed(torch, pair_lib, 3, 1)
    out["rnass_library"] = {"workload": "%d (sequence PFM, structure letter-string PFM) pairs (w=%d), both at -m 6, over the two code streams: "
                                        "k_library on the sequences, the structure letters of its survivors" % (n_lib, args.width),
                            "ms_per_step": ms, "steps": 3, "value": windows * n_lib / (ms * 1e-3), "unit": "window-pair pairs/s",
                            "hits_per_step": int(cnt.item()), "minscore": 6.0, "vs_c5": ms / out["c5"]["ms_per_step"],
                            "vs_per_pair_launches": ms / (n_lib * out["rnass_two_fasta"]["ms_per_step"])}
    plib.close()
    return out

