#!/usr/bin/env python3
"""Compile libhsk for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re
import subprocess
import sys

out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                      "-o", "/tmp/libhsk_res.so", "hysortk_amd/csrc/hsk_api.hip"], stderr=subprocess.PIPE, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
filt = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-60s %5s %5s %8s %5s %8s" % ("kernel", "SGPR", "VGPR", "scratch", "occ", "LDS"))
for r in rows:
    if filt in r["name"]:
        print("%-60s %5s %5s %8s %5s %8s" % (r["name"][-60:], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
