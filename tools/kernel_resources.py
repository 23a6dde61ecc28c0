#!/usr/bin/env python3
"""Register / spill / LDS report of every kernel in csparse3_amd/csrc/kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "csparse3_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=262144", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", "kernels.hip", "-o", "/tmp/cs3_kres.o"]
txt = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for b in re.split(r"Function Name: ", txt)[1:]:
    name = b.split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void cs3::", "")
    if flt not in dn:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    print("%-42s VGPR %4s AGPR %4s SGPR %4s sgpr-spill %4s vgpr-spill %4s scratch %5s waves/SIMD %s LDS %s" % (
        dn[:42], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("SGPRs Spill"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
