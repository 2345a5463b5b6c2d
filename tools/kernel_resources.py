#!/usr/bin/env python3
"""VGPRs, spills, LDS and occupancy of every kernel of the library, from the compiler's own remarks (no GPU needed):
   python tools/kernel_resources.py [pattern]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "hysortk_amd", "csrc", "hsk_api.hip"), "-o", os.path.join(d, "x.o")], capture_output=True, text=True)
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur: rows[cur][k] = v
print("%-64s %6s %7s %7s %8s %8s %4s" % ("kernel", "VGPRs", "sgprSp", "vgprSp", "scratch", "LDS", "occ"))
for name, x in sorted(rows.items()):
    if pat and pat not in name: continue
    print("%-64s %6s %7s %7s %8s %8s %4s" % (name[:64], x.get("VGPRs"), x.get("SGPRs Spill"), x.get("VGPRs Spill"), x.get("ScratchSize [bytes/lane]"), x.get("LDS Size [bytes/block]"), x.get("Occupancy [waves/SIMD]")))
