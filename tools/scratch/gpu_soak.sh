#!/bin/bash
# stability run: many steps at full size, then odd sizes / task counts / K (every run checks the device-side error word
# and the XCD drain checks; a hang would show as a timeout)
export TMPDIR=/tmp
set -e
timeout -k 10 300 python bench.py --steps 40 --warmup 2 --no-cpu | cut -c1-160
for sc in 0.013 0.11 0.37 0.71; do
  for nt in 0 8 24; do
    timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --scale $sc --ntasks $nt | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('scale', d['config']['scale'], 'ntasks', d['config']['ntasks'], round(d['value'] / 1e9, 2), 'G k-mers/s', d['path_stats'])"
  done
done
for k in 21 27 29; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --scale 0.3 --k $k | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('K', d['config']['K'], round(d['value'] / 1e9, 2), 'G k-mers/s', d['path_stats'])"
done
