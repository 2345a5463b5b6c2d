#!/bin/bash
# bench line of another configuration with its kernel table: tools/gpu_cfg_kernels.sh "--ext 1" "--k 51" ...
export TMPDIR=/tmp
for args in "$@"; do
  python bench.py --steps 2 --warmup 1 --no-cpu --no-e2e $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s %.2f G  %.1f ms ' % ('$args', d['value']/1e9, d['ms_per_step']), {k[3:]: round(v,1) for k,v in d['phases_ms_per_step'].items()}, [(k['kernel'][:16], round(k['ms_per_step'],2)) for k in d['kernels']])"
done
