#!/usr/bin/env python3
"""Readable summary of one bench.py JSON line (file given as argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.2f G k-mers/s  %.1f ms/step" % (d["value"] / 1e9, d["ms_per_step"]))
for k in d["kernels"]:
    print("  %-24s %6.2f ms/step  %-14s alg %5.0f GB/s  frac %.3f  of copy %.3f  pmc/launch %s" % (k["kernel"], k["ms_per_step"], k["bound"], k.get("algorithmic_GBs", 0), k.get("frac_of_hbm_peak", 0),
          k.get("frac_of_copy_peak") or 0, k.get("pmc_traffic_bytes_per_launch")))
r = d["roofline"]; print("roofline:", r["kernel"][:48], "achieved %.0f frac %.3f copy_peak %s frac_of_copy %s" % (r["achieved"], r["frac"], r["measured_copy_peak_GBs"], r["frac_of_copy_peak"]))
w = r["whole_path"]; print("whole path:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in w.items() if k in ("GBs", "frac_of_hbm_peak", "frac_of_copy_peak", "reference_algorithm_GBs")})
h = d.get("host_to_host") or {}; print("host_to_host", {k: h.get(k) for k in ("value", "ms_per_step", "h2d_ms", "d2h_ms", "d2h_bytes_per_step", "error")})
for v in d.get("variants", []):
    print("  variant %-15s %s" % (v["name"], {k: (round(v[k], 3) if isinstance(v[k], float) else v[k]) for k in ("value", "ms_per_step", "entries", "error", "reference_algorithm_frac_of_hbm_peak") if k in v}), (v.get("scatter_pass") or {}).get("GBs"))
c = d.get("cpu_baseline", {})
print("cpu", {k: c.get(k) for k in ("kind", "value", "cores", "seconds", "entries", "error")}, c.get("sample_check"))
for l in c.get("layouts", []) or []:
    print("    ", l)
print("    port", (c.get("port") or {}).get("value"))
