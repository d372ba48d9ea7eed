"""Build libpfmscan.so (the HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``rnascan_amd/libpfmscan.so`` travels to the GPU box with the tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpfmscan.so")
SOURCES = ["pfmscan_kernels.hip", "pfmscan_api.hip", "pfmscan_sort.hip", "pfmscan_library.hip", "pfmscan_library_api.hip"]
DEPS = SOURCES + ["pfmscan_internal.hpp", "pfmscan_ctx.hpp", os.path.join("..", "..", "include", "pfmscan.h")]


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


def build_lib(force=False, verbose=False):
    """Compile the library when missing or older than its sources; return its path."""
    if not force and not stale():
        return LIB
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-fno-fast-math", "-Wall",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
