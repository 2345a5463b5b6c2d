#!/bin/bash
# quick regression + bench: sort/count stage tests, golden parity, full-size bench without the cpu leg
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu $QUICK_EXTRA 2>&1 | tail -1 > /tmp/quick_bench.json
python - <<'PY'
import json
d = json.loads(open("/tmp/quick_bench.json").read())
print("value %.3f G k-mers/s  ms/step %.1f" % (d["value"] / 1e9, d["ms_per_step"]))
print({k: round(v, 1) for k, v in d["phases_ms_per_step"].items()})
r = d["roofline"]
print("onesweep %.0f GB/s frac %.3f avg %.3f ms; agg %s GB/s %s ms/launch" % (r["achieved"], r["frac"], r["avg_launch_ms"], r.get("agg_GBs"), r.get("agg_avg_launch_ms")))
PY
