"""Registers, LDS and scratch of every gfx950 kernel in an object file or in libpfmscan.so (no GPU needed).

    python tools/kernel_resources.py [path] [name filter]

Reads the code object's metadata note (`llvm-readelf --notes`): .vgpr_count, .sgpr_count, .group_segment_fixed_size,
.private_segment_fixed_size (scratch), .vgpr_spill_count -- the numbers behind DESIGN.md's "no scratch" statements.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(path):
    """the gfx950 code objects bundled in `path` (a host object or shared library), as temporary files"""
    import shutil
    tmp = tempfile.mkdtemp(prefix="kres_")
    copy = os.path.join(tmp, os.path.basename(path))       # the bundles are extracted next to the input file
    shutil.copy(path, copy)
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], cwd=tmp, stdout=subprocess.DEVNULL)
    return [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "gfx950" in f]


def kernels(co):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out = []
    for block in notes.split("- .agpr_count:")[1:]:
        def grab(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, block)
            return m.group(1) if m else "?"
        name = grab("name")
        try:
            name = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        out.append((name, grab("vgpr_count"), grab("sgpr_count"), grab("group_segment_fixed_size"),
                    grab("private_segment_fixed_size"), grab("vgpr_spill_count"), grab("sgpr_spill_count")))
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "rnascan_amd", "libpfmscan.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%-110s %5s %5s %7s %7s %6s %6s" % ("kernel", "vgpr", "sgpr", "lds", "scratch", "vspill", "sspill"))
    for co in code_objects(path):
        for k in kernels(co):
            if flt in k[0]:
                print("%-110s %5s %5s %7s %7s %6s %6s" % ((re.sub(r"^void pfmscan::", "", k[0])[:110],) + k[1:]))


if __name__ == "__main__":
    main()
