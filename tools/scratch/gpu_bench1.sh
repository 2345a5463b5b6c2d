#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== small bench (scale 0.05)"; timeout 600 python bench.py --scale 0.05 --steps 2 --warmup 1 --cpu-div 20 2>&1 | tail -5
echo "== full bench"; timeout 1200 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -3 gpurun_out/bench_full.err; cat gpurun_out/bench_full.json
nproc; free -g | head -2
