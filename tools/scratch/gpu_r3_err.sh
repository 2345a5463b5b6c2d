#!/bin/bash
# reads with substitution errors: combining extraction (kept on: HSK_COMBINE_RATIO=1) against instance path
export TMPDIR=/tmp
mkdir -p gpurun_out/comberr
for er in ${@:-0.0005 0.001 0.002}; do
  for cfg in HSK_COMBINE_RATIO=1 HSK_COMBINE=0; do
    env $cfg HSK_TIMING=1 timeout -k 10 300 python bench.py --steps 2 --warmup 2 --no-cpu --no-e2e --no-variants --error-rate $er > gpurun_out/comberr/b_${er}_$cfg.json 2> gpurun_out/comberr/b_${er}_$cfg.err || { echo "== $er $cfg FAILED"; tail -3 gpurun_out/comberr/b_${er}_$cfg.err; continue; }
    echo "== error rate $er $cfg"; python tools/bench_summary.py gpurun_out/comberr/b_${er}_$cfg.json 2>/dev/null | head -1; grep "combining" gpurun_out/comberr/b_${er}_$cfg.err | tail -1
  done
done
