#!/usr/bin/env python3
"""Readable summary of one bench.py JSON line (file given as argument)."""
import json, signal, sys
signal.signal(signal.SIGPIPE, signal.SIG_DFL)
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.2f G k-mers/s  %.1f ms/step" % (d["value"] / 1e9, d["ms_per_step"]))
for k in d["kernels"]:
    print("  %-24s %6.2f ms/step  %-14s alg %5.0f GB/s  frac %.3f  of copy %.3f  pmc/launch %s" % (k["kernel"], k["ms_per_step"], k["bound"], k.get("algorithmic_GBs", 0), k.get("frac_of_hbm_peak", 0),
          k.get("frac_of_copy_peak") or 0, k.get("pmc_traffic_bytes_per_launch")))
r = d["roofline"]; print("roofline:", r["kernel"][:48], r["bound"], "achieved %.1f peak %.1f %s frac %.3f floor_ms %s traffic %s" % (r["achieved"], r["peak"], r["unit"][:12], r["frac"], r.get("floor_ms"), r.get("traffic")))
hl = r.get("hbm_leader") or {}; print("  hbm leader:", hl.get("kernel"), "achieved %.0f GB/s frac %.3f traffic %s" % (hl.get("achieved", 0), hl.get("frac", 0), hl.get("traffic")), "copy peak", r.get("measured_copy_peak_GBs"), "pmc same build:", (r.get("traffic_source") or {}).get("same_build"))
for k in d["kernels"]:
    if k.get("valu"): print("  valu %-18s floor %.1f ms of %.1f (%.2f), %.2f cycles/instr, %.3e wave-instr/step" % (k["kernel"], k["valu"]["floor_ms_per_step"], k["ms_per_step"], k["valu"]["frac_of_issue_floor"], k["valu"]["avg_issue_cycles_per_wave_instruction"], k["valu"]["wave_instructions_per_step"]))
w = r["whole_path"]; print("whole path:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in w.items() if k in ("GBs", "frac_of_hbm_peak", "frac_of_copy_peak", "reference_algorithm_GBs")})
h = d.get("host_to_host") or {}; print("host_to_host", {k: h.get(k) for k in ("value", "ms_per_step", "h2d_ms", "d2h_ms", "d2h_bytes_per_step", "error")})
for v in d.get("variants", []):
    print("  variant %-15s %s" % (v["name"], {k: (round(v[k], 3) if isinstance(v[k], float) else v[k]) for k in ("value", "ms_per_step", "first_call_ms", "first_call_device_ms", "first_call_combine_launches", "entries", "error", "reference_algorithm_frac_of_hbm_peak") if k in v}), (v.get("scatter_pass") or {}).get("GBs"))
for f in (d.get("first_calls") or []) if isinstance(d.get("first_calls"), list) else [d.get("first_calls")]:
    print("  first call", f if not isinstance(f, dict) or "input" not in f else {k: (round(v, 2) if isinstance(v, float) else v) for k, v in f.items() if k in ("input", "after_other_input_ms", "steady_ms", "ratio", "combine_launches")})
for m in d.get("multi_rank_path") or []:
    if "error" in m: print("  multi_rank", m); continue
    print("  multi_rank %-16s %5.1f ms per rank (device), %5.1f G k-mers/s per GPU outside the wire, combine launches %s" % (m["name"], m["per_rank_device_ms"]["mean"], m["per_gpu_rate_from_device_ms"] / 1e9, m.get("combine_launches")))
if d.get("large_input"):
    li = d["large_input"]
    print("  large_input", li if "error" in li else "%s: %.1f G k-mers/s, %.0f ms per call, %d tasks, %d entries" % (li["what"], li["value"] / 1e9, li["ms_per_call"], li["ntasks"], li["entries"]))
c = d.get("cpu_baseline", {})
print("cpu", {k: c.get(k) for k in ("kind", "value", "cores", "seconds", "entries", "error")}, c.get("sample_check"))
for l in c.get("layouts", []) or []:
    print("    ", l)
print("    port", (c.get("port") or {}).get("value"))
