#!/bin/bash
# round 3 A/B: tests named in $TESTS (optional), then one short bench line per environment given as arguments ("default" = none)
export TMPDIR=/tmp
TAG=${TAG:-r3ab}
mkdir -p gpurun_out/$TAG
if [ -n "$TESTS" ]; then
  timeout -k 10 800 python -m pytest $TESTS -x -q -m gpu --durations=8 > gpurun_out/$TAG/pytest.log 2>&1 || { tail -30 gpurun_out/$TAG/pytest.log; exit 1; }
  tail -4 gpurun_out/$TAG/pytest.log
fi
i=0
for envs in "$@"; do
  i=$((i+1))
  if [ "$envs" = "default" ]; then e=""; else e="$envs"; fi
  env $e timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu --no-variants $BENCH_ARGS > gpurun_out/$TAG/bench_$i.json 2> gpurun_out/$TAG/bench_$i.err || { echo "FAILED: $envs"; tail -5 gpurun_out/$TAG/bench_$i.err; continue; }
  python3 - "$envs" gpurun_out/$TAG/bench_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
h = d.get("host_to_host") or {}
print("%-34s %.2f G %.1f ms | h2h %.2f G %.1f ms (h2d %.1f d2h %.1f) |" % (sys.argv[1], d["value"] / 1e9, d["ms_per_step"], h.get("value", 0) / 1e9, h.get("ms_per_step", 0), h.get("h2d_ms", 0), h.get("d2h_ms", 0)),
      " ".join("%s %.2f" % (k["kernel"][:8], k["ms_per_step"]) for k in d["kernels"]))
PY
done
