#!/bin/bash
# host wall-clock marks (HSK_TIMING) of the host-to-host leg, one line per environment
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
for cfg in "$@"; do
  if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
  echo "== $cfg"; env $e HSK_TIMING=1 python tools/e2e_probe.py 2>&1 | tail -22 | cut -c1-220
done
