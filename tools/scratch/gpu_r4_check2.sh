#!/bin/bash
O=gpurun_out/${1:-r4o}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_combine.py -x -q -m gpu -k "first_call or alternating or owner_side" > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
for er in 0 0.003; do HSK_TIMING=1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --no-variants --no-e2e --error-rate $er > $O/bench$er.json 2> $O/bench$er.err; grep "plan estimate" $O/bench$er.err | tail -1 | cut -c1-250; python tools/bench_summary.py $O/bench$er.json | head -3; done
