#!/usr/bin/env python3
"""Kernel resource usage (VGPRs, SGPRs, spills, occupancy, LDS) of every gfx950 kernel of
kry_kernels.hip, from the compiler's -Rpass-analysis=kernel-resource-usage remarks.
usage: tools/resource_usage.py [filter-substring ...] > profiles/<round>_kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "krylov_amd", "csrc", "kry_kernels.hip")


def main():
    want = sys.argv[1:] or ["k_fused", "k_final", "k_mgs", "k_bicgn", "k_surf", "k_halo", "g_"]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
           "-Rpass-analysis=kernel-resource-usage", "-c", SRC, "-o", "/dev/null"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    print("%-78s %5s %5s %5s %7s %5s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "LDS"))
    for b, d in zip(blocks, dem):
        d = re.sub(r"\(.*", "", d).replace("void ", "")
        if not any(w in d for w in want):
            continue

        def g(k):
            m = re.search(k + r": (\S+)", b)
            return m.group(1) if m else "?"
        print("%-78s %5s %5s %5s %7s %5s %7s" % (d, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                  g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
