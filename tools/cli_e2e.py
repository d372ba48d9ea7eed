"""End-to-end CLI run at scale on the GPU box: R synthetic records x L nt as a FASTA + a packed profile store, through
bin/rnascan's main() with the real engine; prints wall times per stage.  usage: python tools/cli_e2e.py [R] [L]"""
import io, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

R = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
RBIG = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # a second, sequence-only FASTA of this many records
BIGSTORE = len(sys.argv) > 4 and sys.argv[4] == "store"    # ... with a packed float32 profile store beside it (C3's command line)
from rnascan_amd import cli, store
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
rng = np.random.default_rng(0)
t0 = time.time()
fa = os.path.join(d, "seqs.fa")
letters = np.frombuffer(b"ACGU", dtype=np.uint8)
with open(fa, "wb") as f:
    for i in range(R):
        f.write(b">t%d transcript %d\n" % (i, i))
        f.write(letters[rng.integers(0, 4, size=L)].tobytes() + b"\n")
# the structure letter strings of the same records (SS mode and the two-FASTA RNASS mode, rnascan.py:119-133)
sfa = os.path.join(d, "structs.fa")
sletters = np.frombuffer(b"EHTBLRM", dtype=np.uint8)
with open(sfa, "wb") as f:
    for i in range(R):
        f.write(b">t%d structure %d\n" % (i, i))
        f.write(sletters[rng.integers(0, 7, size=L)].tobytes() + b"\n")
# packed store written directly (the converter's output format), float32
sd = os.path.join(d, "packed")
os.makedirs(sd)
with open(os.path.join(sd, "profile.f32"), "wb") as f:
    for lo in range(0, R, 1000):
        n = min(1000, R - lo)
        p = rng.dirichlet(np.full(7, 0.3), size=n * L).astype(np.float32).reshape(n, L, 7)
        out = np.zeros((n, L + 1, 7), dtype=np.float32)
        out[:, :L] = p
        f.write(out.tobytes())
json.dump({"format": 1, "dtype": "float32", "letters": list("BEHLMRT"), "ids": ["t%d" % i for i in range(R)],
           "lengths": [L] * R, "n_pos": R * (L + 1), "file": "profile.f32", "separator_rows": "one zero row after each record"},
          open(os.path.join(sd, "index.json"), "w"))
big = os.path.join(d, "big.fa")
bigs = os.path.join(d, "big_structs.fa")
if RBIG:
    with open(big, "wb") as f, open(bigs, "wb") as g:
        for lo in range(0, RBIG, 2000):
            n = min(2000, RBIG - lo)
            body = letters[rng.integers(0, 4, size=(n, L))]
            f.write(b"".join(b">t%d transcript %d\n" % (lo + i, lo + i) + body[i].tobytes() + b"\n" for i in range(n)))
            body = sletters[rng.integers(0, 7, size=(n, L))]
            g.write(b"".join(b">t%d structure %d\n" % (lo + i, lo + i) + body[i].tobytes() + b"\n" for i in range(n)))
bigsd = os.path.join(d, "bigpacked")
if RBIG and BIGSTORE:
    os.makedirs(bigsd)
    with open(os.path.join(bigsd, "profile.f32"), "wb") as f:
        for lo in range(0, RBIG, 2000):
            n = min(2000, RBIG - lo)
            p = rng.random((n, L + 1, 7), dtype=np.float32)
            p[p < 0.5] = 0.0                                   # about half exact zeros, like the reference's example profile
            p /= np.maximum(p.sum(axis=2, keepdims=True), 1e-6)
            p[:, L] = 0.0
            f.write(p.tobytes())
    json.dump({"format": 1, "dtype": "float32", "letters": list("BEHLMRT"), "ids": ["t%d" % i for i in range(RBIG)],
               "lengths": [L] * RBIG, "n_pos": RBIG * (L + 1), "file": "profile.f32", "separator_rows": "one zero row after each record"},
              open(os.path.join(bigsd, "index.json"), "w"))
print("inputs written in %.1f s (%s)" % (time.time() - t0, d), file=sys.stderr)
runs = [
    ("seq only  -m 6", ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), "-C", "0.01", "-u", fa]),
    ("seq only  -m -inf (every window a row)", ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), "-C", "0.01", "-u",
                                                "-m", " -inf", fa]),
    ("struct only (store) -m 6", ["-q", os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", sd]),
    ("seq + struct (store) -m 0", ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"),
                                   "-q", os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", "-m", "0",
                                   "--profile-dtype", "float32", fa, sd]),
    ("seq + struct (store) -m -4", ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"),
                                    "-q", os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", "-m", " -4",
                                    "--profile-dtype", "float32", fa, sd]),
]
SPFM, TPFM = os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt")
runs += [("SS: -q pfm structs.fa -m 6", ["-q", TPFM, "-C", "0.01", "-u", sfa]),
         ("SS: -q pfm structs.fa -m 0", ["-q", TPFM, "-C", "0.01", "-u", "-m", "0", sfa]),
         ("SS: -q pfm structs.fa, background from the file", ["-q", TPFM, "-C", "0.01", sfa]),
         ("SS: -q pfm structs.fa -m -inf (every window a row)", ["-q", TPFM, "-C", "0.01", "-u", "-m", " -inf", sfa]),
         ("RNASS two FASTA: -p -q seqs.fa structs.fa -m 0", ["-p", SPFM, "-q", TPFM, "-C", "0.01", "-u", "-m", "0", fa, sfa]),
         ("RNASS two FASTA: -m -6", ["-p", SPFM, "-q", TPFM, "-C", "0.01", "-u", "-m", " -6", fa, sfa])]
if RBIG:
    runs += [("big: SS -q pfm structs.fa -m 6", ["-q", TPFM, "-C", "0.01", "-u", bigs]),
             ("big: SS -q pfm structs.fa -m 0", ["-q", TPFM, "-C", "0.01", "-u", "-m", "0", bigs]),
             ("big: RNASS two FASTA -m 0", ["-p", SPFM, "-q", TPFM, "-C", "0.01", "-u", "-m", "0", big, bigs]),
             ("big: RNASS two FASTA -m -6", ["-p", SPFM, "-q", TPFM, "-C", "0.01", "-u", "-m", " -6", big, bigs])]
if RBIG:
    runs += [("big: seq only -m 6", ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), "-C", "0.01", "-u", big]),
             ("big: seq only -m 2", ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), "-C", "0.01", "-u", "-m", "2", big])]
if RBIG and BIGSTORE:
    for thr in ("0", " -4", " -9", " -12", " -14", " -16"):
        runs.append(("big: seq + struct (store) -m%s" % thr,
                     ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), "-q",
                      os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", "-m", thr,
                      "--profile-dtype", "float32", big, bigsd]))
NDIR = int(os.environ.get("CLI_E2E_DIR", "0"))               # a directory of this many structure.<id>.txt files (the reference's own
if NDIR:                                                     # input format) beside a FASTA of the same records, 1000 nt each
    LD = 1000
    dd = os.path.join(d, "avgdir")
    os.makedirs(dd)
    dfa = os.path.join(d, "dir.fa")
    t1 = time.time()
    with open(dfa, "wb") as f:
        for i in range(NDIR):
            f.write(b">d%d\n" % i + letters[rng.integers(0, 4, size=LD)].tobytes() + b"\n")
            p = rng.dirichlet(np.full(7, 0.3), size=LD)
            with open(os.path.join(dd, "structure.d%d.txt" % i), "w") as g:
                g.write("PO\t" + "\t".join("BEHLMRT") + "\n")
                g.write("".join("%d\t%s\n" % (j, "\t".join(map(repr, row))) for j, row in enumerate(p.tolist())))
    print("directory of %d profile files written in %.1f s" % (NDIR, time.time() - t1), file=sys.stderr)
    for parser in ("native", "pandas"):
        runs.append(("directory (%s parser): struct only, %d files x %d rows" % (parser, NDIR, LD),
                     ["-q", os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", dd]))
        runs.append(("directory (%s parser): seq + struct" % parser,
                     ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), "-q",
                      os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", "-m", "0", dfa, dd]))
NLIB = int(os.environ.get("CLI_E2E_LIBRARY", "0"))          # a library of this many seq+struct PFM pairs (config 5's command line)
if NLIB:
    def write_library(path, letters, seed):
        g = np.random.default_rng(seed)
        with open(path, "w") as f:
            for k in range(NLIB):
                f.write("#RBP%03d\n#PO" % k + "".join("\t" + l for l in letters) + "\n")
                for i, row in enumerate(g.dirichlet(np.full(len(letters), 0.5), size=12)):
                    f.write(str(i) + "".join("\t" + str(float(x)) for x in row) + "\n")
                f.write("\n")
    lib_s, lib_t = os.path.join(d, "seq_lib.pfm"), os.path.join(d, "struct_lib.pfm")
    write_library(lib_s, "ACGU", 1000)
    write_library(lib_t, "EHTBLRM", 2000)
    runs.append(("library: %d pairs, seq + struct (store) -m 6" % NLIB,
                 ["-p", lib_s, "-q", lib_t, "-C", "0.01", "-u", "-m", "6", "--profile-dtype", "float32", fa, sd]))
    runs.append(("library: %d seq PFMs -m 6" % NLIB, ["-p", lib_s, "-C", "0.01", "-u", "-m", "6", fa]))
    # the letter libraries (SURVEY 8f N1 x N4): structure-letter PFMs over a structure FASTA, and pairs over two FASTA files
    runs.append(("library: %d structure-letter PFMs, -q lib structs.fa -m 6" % NLIB, ["-q", lib_t, "-C", "0.01", "-u", "-m", "6", sfa]))
    runs.append(("library: %d pairs, two FASTA -p lib -q lib seqs.fa structs.fa -m 6" % NLIB,
                 ["-p", lib_s, "-q", lib_t, "-C", "0.01", "-u", "-m", "6", fa, sfa]))
    if RBIG:
        runs.append(("big library: %d structure-letter PFMs, -q lib structs.fa -m 6" % NLIB, ["-q", lib_t, "-C", "0.01", "-u", "-m", "6", bigs]))
        runs.append(("big library: %d pairs, two FASTA -m 6" % NLIB, ["-p", lib_s, "-q", lib_t, "-C", "0.01", "-u", "-m", "6", big, bigs]))
    if RBIG and BIGSTORE:
        runs.append(("big library: %d pairs, seq + struct (store) -m 6" % NLIB,
                     ["-p", lib_s, "-q", lib_t, "-C", "0.01", "-u", "-m", "6", "--profile-dtype", "float32", big, bigsd]))
        runs.append(("big library: the same with DEFAULT flags (float32 store scanned as it is)",
                     ["-p", lib_s, "-q", lib_t, "-C", "0.01", "-u", "-m", "6", big, bigsd]))
        runs.append(("big: seq + struct (store) -m -9 with DEFAULT flags",
                     ["-p", os.path.join(DATA, "SLBP_pfm_assembled_normalized_seq.txt"), "-q",
                      os.path.join(DATA, "SLBP_pfm_assembled_normalized_struct.txt"), "-C", "0.01", "-u", "-m", " -9", big, bigsd]))
for name, argv in runs:
    nrec = RBIG if name.startswith("big") else R
    nmot = NLIB if "library" in name else 1
    if name.startswith("directory"):
        os.environ["RNASCAN_PROFILE_PARSER"] = "pandas" if "pandas" in name else "native"
        nrec = NDIR / 3.0                                  # 1000-nt records: a third of the default record in windows
    path = os.path.join(d, "out.tsv")
    with open(path, "w", encoding="utf-8", newline="") as out:
        prof = None
        if os.environ.get("CLI_E2E_PROFILE") and (name.startswith("big") or name.startswith("library") or "-inf" in name):
            import cProfile, pstats
            prof = cProfile.Profile()
            prof.enable()
        t = time.time()
        cli.main(argv, out=out)
        dt = time.time() - t
        if prof is not None:
            prof.disable()
            pstats.Stats(prof, stream=sys.stdout).sort_stats("tottime").print_stats(12)
    rows = sum(1 for _ in open(path, "rb")) - 1
    print("%-52s %.2f s   %d rows   %.3g motif-windows/s   %.1f MB of table" % (name, dt, rows, nmot * nrec * (L - 17) / dt,
                                                                         os.path.getsize(path) / 1e6))
