#!/bin/bash
O=gpurun_out/${1:-r4m}; mkdir -p $O
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-variants --no-e2e --k 51 > $O/bench_base.json 2> $O/bench_base.err
python tools/bench_summary.py $O/bench_base.json | head -7
