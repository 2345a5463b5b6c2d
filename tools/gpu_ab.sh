#!/bin/bash
# A/B of environment switches on the full-size bench: tools/gpu_ab.sh "VAR=a" "VAR=b" ...
export TMPDIR=/tmp
for kv in "$@"; do
  env $kv python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 > /tmp/ab.json
  python - "$kv" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read())
print("%-40s %.3f G k-mers/s  %.1f ms/step  %s ntasks %s %s" % (sys.argv[1], d["value"] / 1e9, d["ms_per_step"], {k[3:]: round(v, 1) for k, v in d["phases_ms_per_step"].items() if k not in ("ms_d2h", "ms_exchange")}, d["config"]["ntasks"], d.get("path_stats")))
PY
done
