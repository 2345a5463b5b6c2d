#!/bin/bash
O=gpurun_out/${1:-r4l}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_combine.py -x -q -m gpu -k "two_word or first_call" > $O/tests_a.log 2>&1; tail -15 $O/tests_a.log
grep -q passed $O/tests_a.log && ! grep -q failed $O/tests_a.log || exit 1
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-variants --no-e2e --k 51 > $O/bench_k51.json 2> $O/bench_k51.err; python tools/bench_summary.py $O/bench_k51.json | head -9
HSK_TUNING=combine=0 timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-variants --no-e2e --k 51 > $O/bench_k51_inst.json 2> $O/bench2.err; python tools/bench_summary.py $O/bench_k51_inst.json | head -8
