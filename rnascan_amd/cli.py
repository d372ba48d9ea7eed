"""``rnascan`` command line on top of the MI355X engine.

Same options, modes, stderr messages, column layout and TSV output as
rnascan/rnascan.py (``getoptions`` :44-105, ``_guess_seq_type`` :114-137,
``main`` :490-576).  Extra options only ADD behaviour: ``--device`` picks the
GPU, ``--pairing`` chooses how averaged-structure columns are paired with the
structure PFM (see scanner.struct_matrix), ``--profile-dtype`` the device storage
of the profiles.  ``-c/--cores`` and ``-x/--debug`` are accepted and
ignored: there is no process pool, a batch is one kernel launch.
"""
import argparse
import os
import sys
import time

import numpy as np

from . import fasta, pack, pssm as pssm_mod, scanner, shard, store, table

from . import __version__


RANK_COMMAND = [sys.executable, "-m", "rnascan_amd"]      # what --gpus N starts once per rank (followed by the arguments)


def getoptions(argv=None):
    desc = "Scan sequence for motif binding sites. Results sent to STDOUT."
    parser = argparse.ArgumentParser(prog="rnascan", description=desc)
    parser.add_argument("fastafiles", metavar="FASTA", nargs="*",
                        help="Input sequence and structure FASTA files")
    pfm_grp = parser.add_argument_group("PFM options")
    pfm_grp.add_argument("-p", "--pfm_seq", dest="pfm_seq", type=str, help="Sequence PFM")
    pfm_grp.add_argument("-q", "--pfm_struct", dest="pfm_struct", type=str, help="Structure PFM")
    parser.add_argument("-C", "--pseudocount", type=float, dest="pseudocount", default=0,
                        help="Pseudocount for normalizing PFM. [%(default)s]")
    parser.add_argument("-m", "--minscore", type=float, dest="minscore", default=6,
                        help="Minimum score for motif hits. [%(default)s]")
    parser.add_argument("-t", "--testseq", dest="testseq", default=None,
                        help=("Supply a test sequence to scan. FASTA files will be ignored. Can supply "
                              "sequence and structure as single string separated by  comma."))
    parser.add_argument("-c", "--cores", type=int, default=8, dest="cores",
                        help="Number of processing cores [%(default)s] (ignored: the scan runs on the GPU)")
    bg_grp = parser.add_argument_group("Background frequency options")
    bg_grp.add_argument("-u", "--uniformbg", action="store_true", default=False, dest="uniform_background",
                        help=("Use uniform background for calculating log-odds [%(default)s]. Default is to "
                              "compute background from input sequences. This option is mutually exclusive with -B."))
    bg_grp.add_argument("-g", "--bgonly", action="store_true", default=False, dest="bgonly",
                        help=("Compute background probabilities from input sequences (STDOUT) and exit. "
                              "[%(default)s]"))
    bg_grp.add_argument("-b", "--bg_seq", default=None, dest="bg_seq",
                        help="Load file of pre-computed background probabilities for nucleotide sequences")
    bg_grp.add_argument("-B", "--bg_struct", default=None, dest="bg_struct",
                        help="Load file of pre-computed background probabilities for nucleotide sequences")
    parser.add_argument("-v", "--version", action="version", version="%(prog)s " + __version__)
    parser.add_argument("-x", "--debug", action="store_true", default=False, dest="debug",
                        help="Turn on debug mode (accepted for compatibility; nothing to disable) [%(default)s]")
    gpu = parser.add_argument_group("MI355X options (not in the reference)")
    gpu.add_argument("--device", type=int, default=int(os.environ.get("RNASCAN_DEVICE", "0")),
                     help="HIP device index [%(default)s]")
    gpu.add_argument("--gpus", type=int, default=None,
                     help=("scan on this many GPUs of the node, one process per GPU, records sharded over them (the analogue "
                           "of the reference's -c: its Pool fan-out).  This process then only starts the ranks and waits; "
                           "rank 0 prints the table [1]"))
    gpu.add_argument("--pairing", choices=["aligned", "positional"], default="aligned",
                     help=("averaged-structure columns vs structure PFM.  'aligned' pairs every profile column with the PFM "
                           "row of the SAME letter -- the evident intent, and what the reference computed on Python 2.  "
                           "NOTE: the reference run on Python >= 3.6 pairs them by POSITION (file order BEHLMRT against the "
                           "PFM's EHTBLRM, rnascan.py:300-307) and therefore prints different structure scores than this "
                           "default; '--pairing positional' reproduces those numbers exactly [%(default)s]"))
    gpu.add_argument("--profile-dtype", choices=["auto", "float64", "float32"], default="float64",
                     help=("device storage of averaged-structure profiles: float64 reproduces the reference's fp64 scores (it "
                           "prints them unrounded, rnascan.py:293-315) to ~1e-14; float32 halves the HBM traffic at a storage "
                           "error of at most 2^-24 x (sum over the PFM's rows of the largest finite |log-odds|); auto takes "
                           "float32 when that bound is below 5e-7 for the structure PFM at hand, else float64, and says so on "
                           "stderr [%(default)s]"))
    args = parser.parse_args(argv)
    if not (args.pfm_seq or args.pfm_struct):
        parser.error("Must specify PFMs with -p and/or -q")
    if args.uniform_background and (args.bg_seq or args.bg_struct):
        parser.error("You cannot set uniform and custom background options at the same time\n")
    return args


def _guess_seq_type(args):
    """RNA, SS or RNASS (rnascan.py:114-137)."""
    nfiles = len(args.fastafiles)
    if nfiles == 2:
        if not (args.pfm_seq or args.pfm_struct):
            fasta.eprint("Missing PFMs")
            sys.exit(1)
        return "RNASS"
    if args.pfm_seq and args.pfm_struct and not args.testseq:
        fasta.eprint("Can't specify two PFMs with one input file")
        sys.exit(1)
    elif args.pfm_seq and args.pfm_struct and args.testseq:
        return "RNASS"
    elif args.pfm_seq:
        return "RNA"
    elif args.pfm_struct:
        return "SS"
    fasta.eprint("Must specify PFMs with -p and/or -q")
    sys.exit(1)


def profile_type(args, struct_pssm, stored=None):
    """the device storage of the profile rows for this run (scanner.pick_profile_dtype), announced once on stderr.
    ``stored``: the dtype of a packed profile store that is the input -- float32 rows ARE the input then: they are scanned
    as they are whatever the PFM's bound says (nothing is rounded on the way, and widening them would change no score)"""
    got = getattr(args, "_profile_type", None)
    if got is None:
        asked = getattr(args, "profile_dtype", "float64")
        if stored is not None and np.dtype(stored) == np.float32 and asked != "float32":
            args._profile_type = np.float32
            fasta.eprint("Averaged-structure profiles are stored as float32 on the device (the packed store holds float32 rows: "
                         "they are the input and are scanned as they are)")
            return np.float32
        got, bound = scanner.pick_profile_dtype(asked, struct_pssm)
        args._profile_type = got
        how = "as asked" if asked != "auto" else \
            ("worst-case storage error %.1e < %.0e" % (bound, scanner.FLOAT32_STORAGE_BUDGET) if got is np.float32 else
             "float32 storage could cost up to %.1e > %.0e" % (bound, scanner.FLOAT32_STORAGE_BUDGET))
        fasta.eprint("Averaged-structure profiles are stored as %s on the device (%s)" % (np.dtype(got).name, how))
    return got


def store_batches(minscore, downcast=False):
    """batch length, in units of RNASCAN_BATCH_POSITIONS, of a scan whose profile rows are slices of a mapped store (nothing
    is copied on the host, the chunked pipeline keeps the device scratch at two chunks): 32 with a finite threshold -- the
    rows of a batch are its hits, and one long pipeline call has one ramp-up instead of one per batch -- 8 at `-m ' -inf'`,
    where every window is a row and a batch also holds its score arrays"""
    import math
    if downcast:                 # float64 store under float32 rows: the batch is copied on the host, keep it at one unit
        return 1
    return 32 if math.isfinite(float(minscore)) else 8


class _Zip(object):
    """two sliceable record sequences of the same length, sliced together (shard.scan_sharded cuts batches with [a:b])"""

    def __init__(self, first, second):
        self.first, self.second = first, second

    def __len__(self):
        return len(self.first)

    def __getitem__(self, key):
        return self.first[key], self.second[key]


def _same_records(seq_fasta, struct_fasta):
    """(LazyFasta of the sequences, LazyFasta of the structures) when the two files hold the same record ids in the same
    order, every id once -- the shape `rnascan seqs.fa structs.fa` is made for: record k of one file pairs with record k
    of the other and a batch of the join is the join of a batch.  Otherwise None: the two tables are made and joined as the
    reference does (rnascan.py:416-434)."""
    recs, srecs = fasta.open_lazy(seq_fasta), fasta.open_lazy(struct_fasta)
    ids = list(recs.ids)
    ok = len(recs) == len(srecs) and len(recs) > 0 and ids == list(srecs.ids) and len(set(ids)) == len(ids)
    return (recs, srecs) if ok else None


def load_motif(pfm_file, pseudocount, letters, background):
    """rnascan.py:210-235 (same messages)."""
    motifs_set = {}
    fasta.eprint("Loading PFM %s" % pfm_file, end="")
    tic = time.time()
    try:
        # a multi-PFM library (pfmutil.py:115-133 format) yields one motif per block; every one is scanned
        motifs_set.update(pssm_mod.load_pssms(pfm_file, pseudocount, letters, background))
    except ValueError:
        fasta.eprint("\nFailed to load motif %s" % pfm_file)
    except KeyError:
        fasta.eprint("\nFailed to load motif %s" % pfm_file)
        fasta.eprint("Check that you are using the correct --type")
        raise
    fasta.eprint("\b.", end="")
    sys.stderr.flush()
    fasta.eprint("done in %0.2f seconds!" % (float(time.time() - tic)))
    fasta.eprint("Found %d motifs" % len(motifs_set))
    if len(motifs_set) == 0:
        raise ValueError("No motifs found.")
    from ._lib import MAX_WIDTH
    wide = [(k, v.length) for k, v in motifs_set.items() if v.length > MAX_WIDTH]
    if wide:
        # the reference's loops take any width (_pwm.c:34-68); here widths up to PFMSCAN_MAX_M run the tuned kernels, wider
        # ones a plain rolled-loop kernel, up to PFMSCAN_MAX_WIDTH
        fasta.eprint("PFM %s in %s is %d positions wide: this build scans PFMs of at most %d positions "
                     "(PFMSCAN_MAX_WIDTH, include/pfmscan.h)" % (wide[0][0], pfm_file, wide[0][1], MAX_WIDTH))
        sys.exit(1)
    return motifs_set


def scan_main(engine, source, pssm, letters, args, dist_ctx=(0, 1, None), sink=None):
    """rnascan.py:335-413: `source` is a FASTA path, a directory of averaged
    structures, or a fasta.Record (the -t test sequence).  With more than one rank
    (dist_ctx = (rank, world, torch.distributed)) every rank scans its contiguous
    share of the records and rank 0 receives the whole table (others get None).  With one rank
    and a ``sink`` the batches' tables go to ``sink(frame)`` one by one and None is returned."""
    rank, world, dist = dist_ctx
    compact = sink is not None                           # streaming: hit columns go to the native writer as they are
    if isinstance(source, fasta.Record):
        df = scanner.scan_records(engine, [source], pssm, letters, args.minscore)
        df["Sequence_ID"] = "testseq"
        df["Description"] = ""
        fasta.eprint("Processed %d sequences" % 1)
        return df
    if store.is_store(source):
        # packed profile store: the mapped file is already the stream the kernel reads
        fasta.eprint("Scanning averaged secondary structures ")
        ps = store.ProfileStore(source)
        ids = list(range(len(ps.ids)))                 # batches of record indices: the mapped file is sliced, not copied
        # the batch is a slice of the mapped file (nothing is copied on the host): batches long enough for the chunked
        # pipeline, which uploads chunk k + 1 beside the scan of chunk k with two chunks of device scratch
        df = shard.scan_sharded(ids, ps.lengths,
                                lambda part: scanner.scan_store(engine, ps, pssm, args.minscore, args.pairing,
                                                                part[0] if part else 0, part[-1] + 1 if part else 0, compact),
                                rank, world, dist, max_positions=store_batches(args.minscore) * shard.batch_positions(), sink=sink)
        fasta.eprint("Processed %d sequences" % len(ps.ids))
        return df
    if os.path.isdir(source):
        fasta.eprint("Scanning averaged secondary structures ")
        files = fasta.list_profiles(source)
        if len(files) == 0:
            raise IOError("No averaged structure files found")
        # the profiles of a batch are parsed when the batch is scanned, so only one batch of them is in memory;
        # ranks and batches are balanced by file size (a text row is ~130 bytes: size / 64 over-estimates the
        # positions, which only makes the batches smaller than they may be)
        weights = [os.path.getsize(path) // 64 + 1 for _, path in files]

        def scan_files(part):
            parsed = fasta.read_profiles([path for _, path in part])           # the files of the batch, on all cores
            named = [(sid, file_letters, prof) for (sid, _), (file_letters, prof) in zip(part, parsed)]
            return scanner.scan_profiles(engine, named, pssm, args.minscore, args.pairing, profile_type(args, pssm), compact)

        df = shard.scan_sharded(files, weights, scan_files, rank, world, dist, sink=sink)
        fasta.eprint("Processed %d sequences" % len(files))
        return df
    fasta.eprint("Scanning sequences ")
    recs = fasta.open_lazy(source)                     # index only: a rank / a batch reads just its own records
    # nucleotide letters only: a position is one byte on the device and four in the score array, so a launch takes 8 x the
    # positions a profile batch may hold
    df = shard.scan_sharded(recs, recs.lengths,
                            lambda part: scanner.scan_records(engine, part, pssm, letters, args.minscore, compact),
                            rank, world, dist, max_positions=(8 if fasta.is_rna_letters(letters) else 1) * shard.batch_positions(), sink=sink)
    fasta.eprint("Processed %d sequences" % len(recs))
    return df


def _init_distributed(args):
    """One process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE): records are sharded over the ranks, there
    is NO data-path collective; the only exchange is the host-side gather of the hit tables onto rank 0, so the
    process group is gloo (RCCL moves nothing here; ``RNASCAN_DIST_BACKEND`` overrides).  Rank r scans on GPU
    LOCAL_RANK.  Returns (rank, world, dist)."""
    rank, world = shard.env_rank_world()
    if world == 1:
        return 0, 1, None
    import torch.distributed as dist
    backend = os.environ.get("RNASCAN_DIST_BACKEND", "gloo")
    if os.environ.get("RNASCAN_ONE_DEVICE") != "1":      # "1": rehearsal of N ranks on a one-GPU box, every rank on --device
        args.device = int(os.environ.get("LOCAL_RANK", str(args.device)))
    if backend == "nccl":
        import torch
        torch.cuda.set_device(args.device)
    if not dist.is_initialized():
        # stdout carries the hit table and nothing else: gloo announces its connections on fd 1 ("[Gloo] Rank 0 is
        # connected to ..."), so the rendezvous (and the first collective, which completes the mesh) runs with fd 1
        # pointing at stderr
        sys.stdout.flush()
        keep = os.dup(1)
        try:
            os.dup2(2, 1)
            dist.init_process_group(backend)
            dist.barrier()
        finally:
            os.dup2(keep, 1)
            os.close(keep)
    return rank, world, dist


def main(argv=None, engine=None, out=None):
    tic = time.time()
    args = getoptions(argv)
    if engine is None and out is None and not args.testseq:
        # --gpus N without an outer launcher: become the parent of N ranks (rnascan.py:388-395 starts Pool(args.cores)
        # at this point).  The parent touches no GPU; rank 0 inherits stdout, so the table goes where it always went.
        from . import launch
        world, must_spawn = launch.resolve_world(args.gpus)
        if must_spawn:
            pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            path = os.pathsep.join([pkg_parent] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p])
            rc, _ = launch.spawn_ranks(world, list(RANK_COMMAND) + list(sys.argv[1:] if argv is None else argv),
                                       extra_env={"PYTHONPATH": path})
            return rc
    out = out or sys.stdout
    seq_type = _guess_seq_type(args)
    testseq_stack = args.testseq.split(",")[::-1] if args.testseq else None
    own_engine = False
    seq_results = struct_results = None
    seq_pssm = struct_pssm = None
    seq_source = struct_source = None

    def get_engine():
        nonlocal engine, own_engine
        if engine is None:
            engine = scanner.HipEngine(args.device)     # raises without libpfmscan / a gfx950 device
            own_engine = True
        return engine

    if seq_type in ("RNA", "RNASS"):
        bg = None
        if args.testseq:
            seq_source = fasta.Record("testseq", "", testseq_stack.pop())
        else:
            seq_source = args.fastafiles[0]
            bg = fasta.load_background(args.bg_seq, args.uniform_background, seq_source, fasta.RNA, not args.bgonly)
        if args.bgonly:
            print(dict(bg), file=out)
            sys.exit()
        seq_pssm = load_motif(args.pfm_seq, args.pseudocount, fasta.RNA, bg)

    if seq_type in ("SS", "RNASS"):
        bg = None
        if args.testseq:
            struct_source = fasta.Record("testseq", "", testseq_stack.pop())
        else:
            struct_source = args.fastafiles[0] if seq_type == "SS" else args.fastafiles[1]
            bg = fasta.load_background(args.bg_struct, args.uniform_background, struct_source, fasta.STRUCT,
                                       not args.bgonly)
        if args.bgonly:
            print(dict(bg), file=out)
            sys.exit()
        struct_pssm = load_motif(args.pfm_struct, args.pseudocount, fasta.STRUCT, bg)

    dist_ctx = _init_distributed(args) if not args.testseq else (0, 1, None)
    rank, world, dist = dist_ctx
    eng = get_engine()
    final = None
    # One rank and one table (no join): every batch is written as soon as it is scanned -- same bytes,
    # Match_ID numbered across batches (rnascan.py:329-332) -- and never held as a whole.
    # Several ranks: rank 0 streams its own rows the same way; every other rank streams its rows, formatted but
    # without Match_ID, into a spool file that rank 0 appends in rank order, numbering as it copies (shard.relay_spools)
    # -- no rank holds a table, the reference's pd.concat of every worker's frames (rnascan.py:407-408) never happens.
    writer = [None]
    spool = [None, None]                                 # (text stream, path) of a rank > 0

    def stream_to(columns):
        if rank == 0:
            writer[0] = table.TsvWriter(out, columns, match_id=True)
        else:
            spool[0], spool[1] = shard.open_spool()
            writer[0] = table.TsvWriter(spool[0], columns, match_id=False, header=False)

        def sink(frame):
            if isinstance(frame, dict):                  # compact hit columns (table.py)
                writer[0].write_chunk(frame)
            elif frame is not None and len(frame):
                writer[0].write_chunk({c: frame[c].to_numpy() for c in writer[0].columns}, len(frame))
        return sink

    streaming = not args.testseq
    pair_files = None
    if seq_type == "RNASS" and not args.testseq and not os.path.isdir(seq_source) and not os.path.isdir(struct_source):
        pair_files = _same_records(seq_source, struct_source)
    if seq_type == "RNASS" and not args.testseq and os.path.isdir(struct_source) and not os.path.isdir(seq_source):
        # sequence FASTA + averaged-structure directory (or packed store): one fused kernel pass per batch (configs 3, 5).
        # Only an index of both sides is held; a batch reads its own records and the profiles of those records.
        ptype = profile_type(args, struct_pssm, store.ProfileStore(struct_source).dtype if store.is_store(struct_source) else None)
        fasta.eprint("Scanning sequences ")
        recs = fasta.open_lazy(seq_source)
        fasta.eprint("Processed %d sequences" % len(recs))
        fasta.eprint("Scanning averaged secondary structures ")
        ps = None
        if store.is_store(struct_source):
            ps = store.ProfileStore(struct_source)
            n_prof = len(ps.ids)
            # the usual store: the FASTA's records in the FASTA's order.  Then every batch is a slice of the mapped file
            # and no per-id index is needed (100k dictionary inserts and look-ups were a tenth of C3's command line)
            rec_ids = list(recs.ids)
            same_order = n_prof == len(recs) and list(ps.ids) == rec_ids and len(set(rec_ids)) == len(rec_ids)
            where = {}
            if not same_order:
                for i, sid in enumerate(ps.ids):
                    where.setdefault(sid, []).append(i)

            def load(sid):
                return [(sid, ps.letters, ps.profile[int(ps.offsets[i]):int(ps.offsets[i] + ps.lengths[i])]) for i in where.get(sid, [])]
        else:
            where = {}
            for sid, path in fasta.list_profiles(struct_source):
                where.setdefault(sid, []).append(path)
            n_prof = sum(len(v) for v in where.values())

            def load(sid):
                out = []
                for path in where.get(sid, []):
                    file_letters, prof = fasta.read_profile(path)
                    out.append((sid, file_letters, prof))
                return out

            def load_many(ids):
                todo = [(sid, path) for sid in ids for path in where.get(sid, [])]
                parsed = fasta.read_profiles([path for _, path in todo])        # the batch's files on all cores
                return [(sid, fl, prof) for (sid, _), (fl, prof) in zip(todo, parsed)]
        if n_prof == 0:
            raise IOError("No averaged structure files found")
        fasta.eprint("Processed %d sequences" % n_prof)
        same_order = ps is not None and same_order
        unique = len(set(recs.ids)) == len(recs) and all(len(v) == 1 for v in where.values())

        def scan_pairs(part):
            """one batch: the fused pass when its records and profiles pair one to one (same id, same length, same
            column order), else the reference's two tables + join for this batch (ids are unique, so the join of a
            batch is the batch of the join)"""
            ids = list(part.ids) if hasattr(part, "ids") else [r.id for r in part]
            named, prepacked = None, None
            if same_order and ids and hasattr(part, "lo"):
                prepacked = (ids, ps.letters, ps.stream(part.lo, part.hi))      # record k of the FASTA is record k of the store
            elif ps is not None and ids:
                # a packed store that holds these records in this order: its rows ARE the stream, no per-record copy
                at = [where.get(rid, [-1])[0] for rid in ids]
                if at[0] >= 0 and at == list(range(at[0], at[0] + len(at))):
                    prepacked = (ids, ps.letters, ps.stream(at[0], at[0] + len(at)))
            if prepacked is None:
                named = load_many(ids) if ps is None else [t for rid in ids for t in load(rid)]
            df = scanner.scan_combined(eng, part, named, seq_pssm, struct_pssm, args.minscore, args.pairing, ptype,
                                       columns=streaming, prepacked=prepacked)
            if df is None:
                if named is None:
                    named = load_many(ids) if ps is None else [t for rid in ids for t in load(rid)]
                df = scanner.combine(scanner.scan_records(eng, part, seq_pssm, fasta.RNA, args.minscore),
                                     scanner.scan_profiles(eng, named, struct_pssm, args.minscore, args.pairing, ptype))
                df = df[scanner.COMBINED_COLUMNS]
            return df

        if unique:
            # a packed store that holds the FASTA's records in the FASTA's order: every batch is a slice of the mapped file
            # plus its packed codes (1 byte per position) -- batches long enough for the chunked upload-beside-scan pipeline
            final = shard.scan_sharded(recs, recs.lengths, scan_pairs, rank, world, dist,
                                       max_positions=(store_batches(args.minscore, ps.profile.dtype == np.float64 and ptype is np.float32)
                                                      if same_order else 1) * shard.batch_positions(),
                                       sink=stream_to(scanner.COMBINED_COLUMNS) if streaming else None)
        else:                                  # duplicate ids join across records: two whole tables + join
            named = load_many(list(where)) if ps is None else [t for sid in where for t in load(sid)]
            seq_results = shard.scan_sharded(
                recs, recs.lengths,
                lambda part: scanner.scan_records(eng, part, seq_pssm, fasta.RNA, args.minscore), rank, world, dist)
            struct_results = shard.scan_sharded(
                named, [p.shape[0] for _, _, p in named],
                lambda part: scanner.scan_profiles(eng, part, struct_pssm, args.minscore, args.pairing, ptype),
                rank, world, dist)
            if rank == 0:
                final = scanner.combine(seq_results, struct_results)
    elif pair_files is not None:
        # sequence FASTA + structure FASTA holding the same records in the same order (rnascan.py:119-123): the letters of
        # both files go to the device as two code streams and a row needs seq > m AND struct > m there (what combine()'s
        # inner join of the two tables keeps, rnascan.py:416-434); rows are streamed batch by batch, no table is held.
        recs, srecs = pair_files
        fasta.eprint("Scanning sequences ")
        fasta.eprint("Processed %d sequences" % len(recs))
        fasta.eprint("Scanning sequences ")
        fasta.eprint("Processed %d sequences" % len(srecs))

        def scan_both(part):
            a, b = part
            df = scanner.scan_pair(eng, a, b, seq_pssm, struct_pssm, args.minscore, columns=streaming)
            if df is None:                     # e.g. a record whose two strings differ in length: this batch's two tables, joined
                df = scanner.combine(scanner.scan_records(eng, a, seq_pssm, fasta.RNA, args.minscore),
                                     scanner.scan_records(eng, b, struct_pssm, fasta.STRUCT, args.minscore))
                df = df[scanner.COMBINED_COLUMNS]
            return df

        final = shard.scan_sharded(_Zip(recs, srecs), recs.lengths, scan_both, rank, world, dist,
                                   max_positions=8 * shard.batch_positions(),
                                   sink=stream_to(scanner.COMBINED_COLUMNS) if streaming else None)
    else:
        one_table = streaming and seq_type in ("RNA", "SS")
        if seq_type in ("RNA", "RNASS"):
            seq_results = scan_main(eng, seq_source, seq_pssm, fasta.RNA, args, dist_ctx,
                                    sink=stream_to(scanner.SEQ_COLUMNS) if one_table else None)
        if seq_type in ("SS", "RNASS"):
            struct_results = scan_main(eng, struct_source, struct_pssm, fasta.STRUCT, args, dist_ctx,
                                       sink=stream_to(scanner.SEQ_COLUMNS) if one_table else None)
        if rank == 0:
            if seq_type == "RNASS":
                final = scanner.combine(seq_results, struct_results)
            elif seq_type == "RNA":
                final = seq_results
            else:
                final = struct_results

    if writer[0] is not None:
        writer[0].close()                        # the last chunk may still be on the writer thread
        if world > 1:
            if spool[0] is not None:
                spool[0].close()
            shard.relay_spools(out, writer[0].rows + 1, spool[1], writer[0].rows if rank else 0, rank, world, dist)
    if rank == 0 and writer[0] is None:
        # Match_ID 1..n after all filtering / joining (rnascan.py:329-332), then the same bytes as
        # DataFrame.to_csv(sep='\t', index=False) (:559-567), written chunk by chunk
        table.write_frame(out, final, match_id=True)
    if own_engine:
        engine.close()
    if dist is not None:
        dist.barrier()
    runtime = float(time.time() - tic)
    if runtime > 60:
        fasta.eprint("Done in %0.4f minutes!" % (runtime / 60))
    else:
        fasta.eprint("Done in %0.4f seconds!" % (runtime))
    return 0


if __name__ == "__main__":
    sys.exit(main())
