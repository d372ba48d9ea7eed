"""Build libpfmscan.so (the HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``rnascan_amd/libpfmscan.so`` travels to the GPU box with the tree.
Every source is compiled to its own object (in parallel, only when it or a header
changed), then linked.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libpfmscan.so")
SOURCES = ["pfmscan_kernels.hip", "pfmscan_letters8.hip", "pfmscan_letters_fixed.hip", "pfmscan_api.hip", "pfmscan_sort.hip", "pfmscan_library.hip", "pfmscan_library_api.hip", "pfmscan_proflib.hip", "pfmscan_profile_fixed.hip", "pfmscan_place.hip",
           "pfmscan_pipeline.hip", "pfmscan_ingest.hip", "pfmscan_upload.hip"]
HEADERS = ["pfmscan_internal.hpp", "pfmscan_ctx.hpp", "pfmscan_device.hpp", "pfmscan_profile.hpp", "pfmscan_exact.hpp", os.path.join("..", "..", "include", "pfmscan.h")]
DEPS = SOURCES + HEADERS
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wall"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def _obj_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in [src] + HEADERS)


def build_lib(force=False, verbose=False):
    """Compile the library when missing or older than its sources; return its path."""
    if not force and not stale():
        return LIB
    hipcc = hipcc_path()
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        if force or _obj_stale(src, obj):
            jobs.append([hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + \
           [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    run(link)
    bad = runtime_indexed_registers(LIB)
    if bad:
        os.remove(LIB)
        raise RuntimeError("kernels index registers at run time (see runtime_indexed_registers): %r" % bad)
    return LIB


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
# instructions that index the register file at run time (VGPR index mode, relative moves)
_RUNTIME_INDEXED = re.compile(r"\b(s_set_gpr_idx_on|s_set_gpr_idx_idx|v_movrel[sd]+_b32|s_movrel[sd]_b(?:32|64))\b")


def runtime_indexed_registers(path=LIB):
    """Kernels of the gfx950 code objects in `path` that index registers at RUN TIME -> {kernel symbol: count}.

    None of this library's kernels means to: every register array is indexed by constant expressions.  When one is not
    (a loop the unroller left rolled), hipcc promotes the array to a register tuple, if-converts a guarded update
    `if (0 <= u && u <= W) pk[u] += x` into an unconditional `s_set_gpr_idx_on u, gpr_idx(DST)` / `v_mov_b32` write and
    does not clamp u: out-of-range indices overwrite unrelated live registers.  That was the wrong result of
    k_letters_cred8<16> in round 4 (profiles/r5/NOTES.md; tools/gpr_idx_oob.hip isolates it), so a library in which the
    pattern appears is refused."""
    tmp = tempfile.mkdtemp(prefix="pfmscan_isa_")
    try:
        copy = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, copy)
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", copy], cwd=tmp, stdout=subprocess.DEVNULL)
        found = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            asm = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)],
                                 capture_output=True, text=True, check=True).stdout
            sym = "?"
            for line in asm.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    sym = m.group(1)
                elif _RUNTIME_INDEXED.search(line):
                    found[sym] = found.get(sym, 0) + 1
        return found
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def build_floor_tool(force=False):
    """tools/hbm_mixed: the microbenchmark that moves the headline kernel's byte mix without scoring (bench.py runs its
    `quick` form beside the headline so that `roofline.mixed_read_write_floor` is a measurement of the SAME box)."""
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    src, exe = os.path.join(tools, "hbm_mixed.hip"), os.path.join(tools, "hbm_mixed")
    if force or not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call([hipcc_path(), "-O3", "--offload-arch=gfx950", src, "-o", exe, "-ldl"])
    return exe


if __name__ == "__main__":
    build_floor_tool(force="--force" in sys.argv)
    print(build_lib(force="--force" in sys.argv, verbose=True))
