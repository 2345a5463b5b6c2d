#!/bin/bash
# A/B of environment switches on the bench workload: tools/gpu_env_ab.sh "VAR=V VAR2=V" "VAR=W" ...   ("-" = defaults)
export TMPDIR=/tmp
for setting in "$@"; do
  ( if [ "$setting" != "-" ]; then export $setting; fi
    python bench.py --steps 3 --warmup 1 --no-cpu --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-32s %.2f G  %.1f ms ' % ('$setting', d['value']/1e9, d['ms_per_step']), [(k['kernel'][:14], round(k['ms_per_step'],2)) for k in d['kernels']])" )
done
