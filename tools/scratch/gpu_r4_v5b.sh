#!/bin/bash
O=gpurun_out/${1:-r4h}; mkdir -p $O
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-variants --no-e2e > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json | head -8
