#!/usr/bin/env python3
"""cProfile of the host path of a sequence scan (what `rnascan -p pfm.txt big.fa` does): where the wall time goes"""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, ".")
from rnascan_amd import cli

def main():
    n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 3000
    path = "/tmp/cli_profile.fa"
    rng = np.random.default_rng(0)
    t = time.time()
    lut = np.frombuffer(b"ACGU", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(n):
            s = lut[rng.integers(0, 4, L)]
            f.write(b">r%d some description\n" % i)
            f.write(b"\n".join(bytes(s[k:k + 60]) for k in range(0, L, 60)) + b"\n")
    print("wrote %s in %.1f s" % (path, time.time() - t), file=sys.stderr)
    pfm = os.path.join("tests", "golden", "data", "SLBP_pfm_assembled_normalized_seq.txt")
    out = io.StringIO()
    pr = cProfile.Profile()
    t = time.time()
    pr.enable()
    cli.main(["-p", pfm, "-u", "-m", "4", path], out=out)
    pr.disable()
    print("cli.main: %.2f s, %d output lines" % (time.time() - t, out.getvalue().count("\n")), file=sys.stderr)
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("cumulative").print_stats(28)

main()
