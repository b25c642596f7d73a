#!/usr/bin/env python3
"""Print registers / scratch / LDS of the kernels of a built library (from the code object's metadata notes).

    python3 tools/kernel_meta.py [lib.so] [name filter]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(root, "groth_sahai_rs_amd", "lib", "libgs_amd.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co, "--unbundle"])
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
        syms = subprocess.check_output([LLVM + "/llvm-readelf", "-s", "-W", co], text=True)
    size = {}
    for ln in syms.splitlines():
        p = ln.split()
        if len(p) == 8 and p[3] == "FUNC":
            size[p[7]] = int(p[2])
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        try:
            dem = subprocess.check_output(["c++filt", name], text=True).strip()
        except Exception:
            dem = name
        if flt and flt not in dem:
            continue
        short = re.sub(r"\(.*", "", dem).replace("void ", "")
        print("%-60s agpr %3s vgpr %3s sgpr %3s scratch %6s lds %6s code %7d B" % (
            short[:60], re.match(r"\s*(\d+)", blk).group(1), g("vgpr_count").group(1), g("sgpr_count").group(1),
            g("private_segment_fixed_size").group(1), g("group_segment_fixed_size").group(1), size.get(name, 0)))


if __name__ == "__main__":
    main()
