#!/bin/bash
# informational bench lines for the other BASELINE configs on one GPU: K=51 (two-word keys), K=31 EXTENSION=1
export TMPDIR=/tmp
for args in "--k 51" "--ext 1" "$@"; do
  [ -z "$args" ] && continue
  python bench.py --steps 2 --warmup 1 --no-cpu --no-e2e $args 2>/dev/null | tail -1 > /tmp/cfg.json
  python - "$args" <<'PY'
import json, sys
d = json.loads(open("/tmp/cfg.json").read())
r = d["roofline"]
print("%-24s %.3f G k-mers/s  %.1f ms/step  %s  onesweep %.0f GB/s x %d launches of %.2f ms  whole-path %.0f GB/s" % (sys.argv[1], d["value"] / 1e9, d["ms_per_step"],
      {k[3:]: round(v, 1) for k, v in d["phases_ms_per_step"].items() if k not in ("ms_d2h", "ms_exchange")}, r["achieved"], r["launches"], r["avg_launch_ms"], r["whole_path"]["reference_algorithm_GBs"]))
PY
done
