#!/bin/bash
export TMPDIR=/tmp
VAR=${1:-HSK_HYBRID}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -3
for mode in 0 1; do
echo "$VAR=$mode"
env $VAR=$mode timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.3f G k-mers/s  ms/step %.1f ntasks %d' % (d['value']/1e9, d['ms_per_step'], d['config']['ntasks']))
print({k: round(v,1) for k,v in d['phases_ms_per_step'].items()})
r=d['roofline']; print('onesweep %.0f GB/s frac %.3f avg %.3f ms launches %d' % (r['achieved'], r['frac'], r['avg_launch_ms'], r['launches']))"
done
