#!/bin/bash
# bench at several task counts: tools/gpu_ntasks.sh 24 32 40 ...
export TMPDIR=/tmp
for nt in "$@"; do
  python bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --ntasks $nt 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$nt" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read())
print("ntasks %-4s %.3f G k-mers/s  %.1f ms/step  %s %s" % (sys.argv[1], d["value"] / 1e9, d["ms_per_step"], {k[3:]: round(v, 1) for k, v in d["phases_ms_per_step"].items() if k not in ("ms_d2h", "ms_exchange")}, d.get("path_stats")))
PY
done
