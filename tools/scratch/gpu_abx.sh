#!/bin/bash
# A/B of library builds on the fused kernel only (results may be invalid for diagnostic builds): prints extract ms
export TMPDIR=/tmp
for lib in "$@"; do
  HSK_LIB=$lib python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 > /tmp/ab.json
  python - "$lib" <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/ab.json").read())
    print("%-40s %.1f ms/step  %s" % (sys.argv[1], d["ms_per_step"], {k[3:]: round(v, 1) for k, v in d["phases_ms_per_step"].items() if k not in ("ms_d2h", "ms_exchange")}))
except Exception as e:
    print(sys.argv[1], "failed:", open("/tmp/ab.json").read()[:300])
PY
done
