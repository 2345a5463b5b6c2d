#!/usr/bin/env python3
"""profiles/pmc.json from the rocprofv3 --pmc passes of tools/gpu_profile_round.sh (one counter set per pass, kernel trace only).

  python3 tools/pmc_to_json.py <dir with the passes> <out.json> <steps the traced command ran (warm-up included)> <bench line of one pass>

Per kernel (template arguments kept, `void hsk::` stripped), PER STEP of the traced bench command: launches and every counter summed
over the step's dispatches.  HBM bytes: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a
streaming read, i.e. half the bytes (MI355X_MICROARCH.md, HBM section) -- `hbm_bytes_per_step` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.
bench.py reads this file for `traffic` and for the VALU-bound kernels' instruction counts; `build_sha16` (sha256 of libhsk.so) says
which build the passes ran on."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

root, out, steps = sys.argv[1], sys.argv[2], max(int(sys.argv[3]), 1)
bench_line = sys.argv[4] if len(sys.argv) > 4 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hysortk_amd.build import source_sha16  # noqa: E402
agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void hsk::", "").replace("hsk::", "")
        k = k.split("(")[0].strip()
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k][row["Counter_Name"]].add(row["Dispatch_Id"])
kern = {}
skip = ("synth_", "copy_peak_kernel", "__amd_rocclr", "at::native")
path_bytes = 0.0
for k in sorted(agg):
    n = max(len(s) for s in disp[k].values())
    d = {"launches_per_step": n / steps}
    for cn, v in agg[k].items():
        d[cn + "_per_step"] = v / steps
    if "FETCH_SIZE" in agg[k] or "WRITE_SIZE" in agg[k]:
        d["hbm_bytes_per_step"] = (2 * agg[k].get("FETCH_SIZE", 0.0) + agg[k].get("WRITE_SIZE", 0.0)) * 1024 / steps
        if not k.startswith(skip):
            path_bytes += d["hbm_bytes_per_step"]
    kern[k] = d
lib = os.path.join(ROOT, "hysortk_amd", "libhsk.so")
o = {"what": "rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-variants` (full BASELINE configs[1] workload), per step",
     "units": "FETCH_SIZE / WRITE_SIZE in KB; hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half of a streamed read)",
     "build_sha16": source_sha16(), "build_sha16_is": "sha256 over hysortk_amd/csrc/*.h, *.hip (hysortk_amd/build.py source_sha16): the sources this run's library was built from",
     "steps_traced": steps, "path_hbm_bytes_per_step": path_bytes, "kernels": kern}
if bench_line and os.path.exists(bench_line):
    try:
        b = json.loads(open(bench_line).read().strip().splitlines()[-1])
        o["workload"] = {"config": b["config"]["workload"], "scale": b["config"]["scale"], "plan": b.get("plan"), "kmers_per_step": b["value"] * b["ms_per_step"] * 1e-3}
    except Exception as e:
        o["workload_error"] = str(e)
json.dump(o, open(out, "w"), indent=1)
top = sorted(((d.get("SQ_INSTS_VALU_per_step", 0), k) for k, d in kern.items()), reverse=True)[:8]
for v, k in top:
    d = kern[k]
    print("%-60s launches/step %6.1f  VALU %.3e  SALU %.3e  LDS %.3e  hbm %.3e B" % (k[:60], d["launches_per_step"], v, d.get("SQ_INSTS_SALU_per_step", 0), d.get("SQ_INSTS_LDS_per_step", 0), d.get("hbm_bytes_per_step", 0)))
print("path hbm bytes per step %.4e" % path_bytes)
