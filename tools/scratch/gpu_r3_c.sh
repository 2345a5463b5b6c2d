#!/bin/bash
# round 3: new full-size digest tests + the full default bench line
export TMPDIR=/tmp
TAG=${1:-r3c}
mkdir -p gpurun_out/$TAG
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size" --durations=10 > gpurun_out/$TAG/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/$TAG/pytest.log
[ $rc -ne 0 ] && exit $rc
SECONDS=0; timeout -k 10 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rc=$?
echo "bench.py wall: ${SECONDS}s rc=$rc"; tail -5 gpurun_out/$TAG/bench.err | cut -c1-300
python3 - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print("value %.2f G k-mers/s  %.1f ms/step" % (d["value"] / 1e9, d["ms_per_step"]))
for k in d["kernels"]:
    print("  %-24s %6.2f ms/step  %s  alg %.0f GB/s  pmc %s" % (k["kernel"], k["ms_per_step"], k["bound"], k.get("algorithmic_GBs", 0), k.get("pmc_traffic_bytes_per_launch")))
r = d["roofline"]; print("roofline", r["kernel"][:40], r["achieved"], r["frac"], r["measured_copy_peak_GBs"], r["frac_of_copy_peak"])
h = d.get("host_to_host", {}); print("h2h", {k: h.get(k) for k in ("value", "ms_per_step", "h2d_ms", "d2h_ms", "error")})
for v in d.get("variants", []):
    print("  variant %-16s %s" % (v["name"], {k: (round(v[k], 3) if isinstance(v[k], float) else v[k]) for k in ("value", "ms_per_step", "entries", "error") if k in v}), v.get("scatter_pass"), v.get("reference_algorithm_frac_of_hbm_peak"))
c = d.get("cpu_baseline", {})
print("cpu", {k: c.get(k) for k in ("kind", "value", "cores", "seconds", "entries", "error")}, c.get("sample_check"), c.get("layouts"), (c.get("port") or {}).get("value"))
PY
exit $rc
