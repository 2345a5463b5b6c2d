#!/bin/bash
# selected tests (arguments after the tag) + the full default bench line (e2e_host and cpu_baseline legs included)
export TMPDIR=/tmp
TAG=${1:-bf}; shift
mkdir -p gpurun_out/$TAG
if [ $# -gt 0 ]; then
  timeout -k 10 600 python -m pytest "$@" -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1 || { tail -30 gpurun_out/$TAG/pytest.log; exit 1; }
  tail -3 gpurun_out/$TAG/pytest.log
fi
SECONDS=0; timeout -k 10 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rc=$?
echo "bench.py wall: ${SECONDS}s"; tail -5 gpurun_out/$TAG/bench.err | cut -c1-300
python3 - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print("value %.2f G k-mers/s  %.1f ms/step  syncs %.1f (covered %.1f)" % (d["value"] / 1e9, d["ms_per_step"], d["host_syncs_per_step"], d["host_waits_covered_per_step"]))
print({k: round(v, 1) for k, v in d["phases_ms_per_step"].items()})
for k in d["kernels"]:
    print("  %-24s %6.2f ms/step  %s  alg %.0f GB/s" % (k["kernel"], k["ms_per_step"], k["bound"], k.get("algorithmic_GBs", 0)))
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "measured_copy_peak_GBs", "frac_of_copy_peak")}, d["roofline"]["whole_path"]["GBs"], d["roofline"]["whole_path"]["frac_of_hbm_peak"])
print("e2e", d.get("e2e_host"))
c = d.get("cpu_baseline", {})
print("cpu", {k: c.get(k) for k in ("kind", "value", "cores", "seconds", "entries", "sample_fraction")}, c.get("layouts"), (c.get("port") or {}).get("value"))
PY
exit $rc
