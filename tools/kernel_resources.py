#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy summary of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py synfmc_amd/csrc/gemm_conv.hip [regex]"""
import re
import subprocess
import sys

src, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-fno-honor-nans" if "spatial_attn.hip" in src else "-O3",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs Spill|VGPRs|AGPRs|ScratchSize|Occupancy)[^:]*: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    if re.search(filt, name):
        print("%-64s vgpr %4d agpr %4d scratch %5d vspill %4d occ %d" % (name[:64], v.get("VGPRs", 0), v.get("AGPRs", 0), v.get("ScratchSize", 0),
                                                                          v.get("VGPRs Spill", 0), v.get("Occupancy", 0)))
