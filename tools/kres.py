#!/usr/bin/env python
"""Register / LDS / occupancy table of our own kernels in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), library kernels filtered out.
   python tools/kres.py salmon_amd/csrc/hip/map.hip"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-x", "hip",
                    "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None; rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
    if m and cur: rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
for k, v in rows.items():
    if "rocprim" in k or "hipcub" in k: continue
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-40s VGPR %3d  SGPR %3d  scratch %4d  LDS %6d  occ %d" % (name[-40:], v.get("VGPRs", -1), v.get("TotalSGPRs", -1), v.get("ScratchSize", -1), v.get("LDS", -1), v.get("Occupancy", -1)))
