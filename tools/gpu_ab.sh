#!/bin/bash
# A/B in one box: single-task vs XCD-batched sort
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for mode in 0 1; do
echo "HSK_XCD_BATCH=$mode"
HSK_XCD_BATCH=$mode timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --ntasks 40 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.3f G k-mers/s  ms/step %.1f' % (d['value']/1e9, d['ms_per_step']))
print({k: round(v,1) for k,v in d['phases_ms_per_step'].items()})
r=d['roofline']; print('onesweep %.0f GB/s frac %.3f avg %.3f ms launches %d' % (r['achieved'], r['frac'], r['avg_launch_ms'], r['launches']))"
done
