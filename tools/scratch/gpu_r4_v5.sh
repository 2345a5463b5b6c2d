#!/bin/bash
O=gpurun_out/${1:-r4g}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_combine.py -x -q -m gpu > $O/tests_a.log 2>&1; tail -12 $O/tests_a.log
grep -q passed $O/tests_a.log && ! grep -q failed $O/tests_a.log || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-variants > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json | head -16
HSK_TUNING=scan_place=0 timeout -k 10 600 python bench.py --steps 5 --warmup 1 --no-cpu --no-variants --no-e2e > $O/bench_noscanplace.json 2> $O/bench2.err; python tools/bench_summary.py $O/bench_noscanplace.json | head -10
