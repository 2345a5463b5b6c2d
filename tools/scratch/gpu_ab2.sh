export TMPDIR=/tmp
for kv in HSK_UNSTABLE_FIRST=0 HSK_UNSTABLE_FIRST=1; do for args in "--k 51" "--ext 1"; do
  env $kv python bench.py --steps 2 --warmup 1 --no-cpu $args 2>&1 | tail -1 > /tmp/ab.json
  python - "$kv $args" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read())
print("%-34s %.3f G  %.1f ms  sort %.1f" % (sys.argv[1], d["value"] / 1e9, d["ms_per_step"], d["phases_ms_per_step"]["ms_sort"]))
PY
done; done
