#!/usr/bin/env python3
"""tools/kernel_resources.py -- registers, spills, scratch and LDS of every kernel in the library, from the compiler's own metadata
(hipcc -S --cuda-device-only per source file; no GPU needed).  Writes a table: profiles/rNN_resources.txt.
  python tools/kernel_resources.py > profiles/r05_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "kvazzup_amd", "csrc")
FILES = ["enc_kernels.hip", "dec_kernels.hip", "subpel_kernels.hip", "cabac_kernels.hip", "rc_kernels.hip", "color_kernels.hip", "harness_kernels.hip"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    rows = []
    for f in FILES:
        with tempfile.TemporaryDirectory() as td:
            s = os.path.join(td, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", s, os.path.join(SRC, f)],
                           check=True, stderr=subprocess.DEVNULL, cwd=SRC)
            text = open(s).read()
        # the metadata block: one YAML-ish entry per kernel
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
            e = m.group(0)
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, e) or [None, "?"])[1]
            rows.append((f, g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    names = demangle([r[1] for r in rows])
    print("# registers, spills, scratch (private segment) and LDS per kernel of libkvazzup_amd.so -- compiler metadata, hipcc -O3 --offload-arch=gfx950 (tools/kernel_resources.py)")
    print("%-18s %-78s %5s %5s %5s %10s %10s %8s %7s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "sgpr_spill", "vgpr_spill", "scratch", "lds"))
    for r in rows:
        n = names[r[1]].replace("kvzx::", "").split("(")[0].replace("void ", "")
        print("%-18s %-78s %5s %5s %5s %10s %10s %8s %7s" % (r[0], n[:78], r[2], r[3], r[4], r[5], r[6], r[7], r[8]))


if __name__ == "__main__":
    main()
