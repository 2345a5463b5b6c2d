#!/bin/bash
# A/B of alternative builds on another configuration: tools/gpu_ab_cfg.sh "<bench args>" <lib-or-default> ...
export TMPDIR=/tmp
ARGS=$1; shift
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset HSK_LIB; else export HSK_LIB=$PWD/$lib; fi
  python bench.py --steps 2 --warmup 1 --no-cpu --no-e2e $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %.2f G  %.1f ms ' % ('$lib', d['value']/1e9, d['ms_per_step']), [(k['kernel'][:14], round(k['ms_per_step'],2)) for k in d['kernels']])"
done
