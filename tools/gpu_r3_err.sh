#!/bin/bash
# reads with substitution errors: combining extraction against instance path
export TMPDIR=/tmp
mkdir -p gpurun_out/comberr
for er in 0.003 0.01; do
  for cfg in default HSK_COMBINE=0; do
    if [ "$cfg" = "default" ]; then e=""; else e="$cfg"; fi
    env $e HSK_TIMING=1 timeout -k 10 300 python bench.py --steps 2 --warmup 2 --no-cpu --no-e2e --no-variants --error-rate $er > gpurun_out/comberr/b_${er}_$cfg.json 2> gpurun_out/comberr/b_${er}_$cfg.err || { echo "== $er $cfg FAILED"; tail -3 gpurun_out/comberr/b_${er}_$cfg.err; continue; }
    echo "== error rate $er $cfg"; python tools/bench_summary.py gpurun_out/comberr/b_${er}_$cfg.json | head -1; grep "combining" gpurun_out/comberr/b_${er}_$cfg.err | tail -1
  done
done
