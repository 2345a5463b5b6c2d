#!/bin/bash
for nt in 38 128 256 512 1024; do
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu --ntasks $nt 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('ntasks %d: %.3f G k-mers/s' % (d['config']['ntasks'], d['value']/1e9), {k: round(v,1) for k,v in d['phases_ms_per_step'].items()}, 'onesweep %.0f GB/s avg %.3f ms' % (r['achieved'], r['avg_launch_ms']))"
done
