#!/bin/bash
# informational: the benchmark workload with 1 % substitution errors (aggregation table ladder under real-data conditions)
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "errors" 2>&1 | tail -3
for er in 0.0 0.01; do
  python bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --error-rate $er 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('error_rate', d['config']['error_rate'], 'value %.2f G k-mers/s %.1f ms/step' % (d['value']/1e9, d['ms_per_step']), d['path_stats'], {k: round(v,1) for k,v in d['phases_ms_per_step'].items()}, d['config']['output'])"
done
