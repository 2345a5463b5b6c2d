#!/bin/bash
O=gpurun_out/${1:-r4n}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_shim_cpp.py tests/test_gpu_combine.py -x -q -m gpu -s > $O/tests_a.log 2>&1; grep -E "device ingest|passed|failed" $O/tests_a.log | tail -5
grep -q failed $O/tests_a.log && { tail -30 $O/tests_a.log; exit 1; }
HSK_TIMING=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-variants --no-e2e > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json | head -9; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('first call in process', d.get('first_call_in_process'))"
