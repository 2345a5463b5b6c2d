#!/usr/bin/env python3
"""Instruction mix of one kernel from the gfx950 assembly (hipcc -S --cuda-device-only)."""
import re, subprocess, sys
from collections import Counter
src = "hysortk_amd/csrc/hsk_api.hip"
subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "/tmp/hsk.s", src], stderr=subprocess.DEVNULL)
lines = open("/tmp/hsk.s").read().split("\n")
pat = sys.argv[1]
start = [i for i, l in enumerate(lines) if re.match(r"^_ZN3hsk.*:", l) and pat in l]
for st in start:
    name = lines[st].split(":")[0]
    ins = []
    for l in lines[st + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        t = l.strip()
        if l.startswith("\t") and t and not t.startswith((".", ";")):
            ins.append(t.split()[0])
    c = Counter(ins)
    print(name, "instructions:", len(ins))
    groups = Counter()
    for k, v in c.items():
        g = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_")) else "other"
        groups[g] += v
    print("  groups:", dict(groups))
    print("  top:", c.most_common(30))
