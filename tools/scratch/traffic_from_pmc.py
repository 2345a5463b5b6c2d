#!/usr/bin/env python3
"""profiles/traffic.json from a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run (tools/gpu_pmc.sh, full scale):
HBM bytes per launch of the dominant kernel.  gfx950: FETCH_SIZE counts 64 B per 128-B request, i.e. exactly
half of a coalesced streaming read (MI355X_MICROARCH.md, HBM); hist_kernel (reads exactly n*8 B with the same
8-B-per-lane pattern) is used as the in-situ calibration of that factor."""
import csv, glob, json, os, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void hsk::", "").replace("hsk::", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
res = {}
for k in agg:
    n = max(len(calls[k].get("FETCH_SIZE", [])), len(calls[k].get("WRITE_SIZE", [])), 1)
    res[k] = {"launches": n, "FETCH_SIZE_KB_per_launch": agg[k].get("FETCH_SIZE", 0) / n, "WRITE_SIZE_KB_per_launch": agg[k].get("WRITE_SIZE", 0) / n}
dom = [k for k in res if k.startswith("onesweep_multi_kernel")] or [k for k in res if k.startswith("onesweep_kernel")]
o = {"units": "rocprofv3 FETCH_SIZE/WRITE_SIZE are KB; bytes = KB*1024; read side doubled (gfx950 FETCH_SIZE = 1/2 of streamed bytes)", "kernels": res}
if dom:
    d = res[dom[0]]
    o["dominant_kernel"] = dom[0]
    o["onesweep_bytes_per_launch"] = (2 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024
# whole counting path (one step; the PMC passes run bench.py --steps 1 --warmup 0): everything but the synthetic-read generator
path = 0.0
for k, d in res.items():
    # not part of the counting path: the synthetic-read generator, and bench.py's own device-to-device copies (the measured
    # copy peak: torch copies of 2 x 4 GiB run as __amd_rocclr_copyBuffer / at::native kernels; the path's own copyBuffer
    # traffic, a 0.5 GB read-index copy per step, is dropped with them)
    if k.startswith("synth_") or "at::native" in k or k.startswith("__amd_rocclr_copyBuffer") or k.startswith("copy_peak_kernel"):
        continue
    path += d["launches"] * (2 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024
o["path_bytes_per_step"] = path
json.dump(o, open(out, "w"), indent=1)
print(json.dumps({k: o[k] for k in o if k != "kernels"}, indent=1))
