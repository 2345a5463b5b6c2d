#!/bin/bash
# bench (device-resident leg only) under a list of environment settings; one summary line each
export TMPDIR=/tmp
mkdir -p gpurun_out/combenv
i=0
for cfg in "$@"; do
  i=$((i+1))
  if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
  env $e timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-variants > gpurun_out/combenv/b$i.json 2> gpurun_out/combenv/b$i.err || { echo "== $cfg FAILED"; tail -3 gpurun_out/combenv/b$i.err; continue; }
  echo "== $cfg"; python tools/bench_summary.py gpurun_out/combenv/b$i.json | head -7 | cut -c1-110
done
