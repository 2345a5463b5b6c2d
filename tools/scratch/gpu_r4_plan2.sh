#!/bin/bash
O=gpurun_out/${1:-r4d}; mkdir -p $O
./tools/exp/malloc_cost > $O/malloc_cost.txt 2>&1; cat $O/malloc_cost.txt
python -m pytest tests/test_gpu_combine.py -x -q -m gpu -k "first_call or alternating" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for er in 0 0.003 0.75; do
  HSK_TIMING=1 python bench.py --steps 3 --warmup 1 --no-cpu --no-variants --no-e2e --error-rate $er > $O/bench_er$er.json 2> $O/bench_er$er.err
  grep "plan estimate" $O/bench_er$er.err | tail -1; python tools/bench_summary.py $O/bench_er$er.json | head -3
done
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "full_size or more_than_eight" > $O/tests_mr.log 2>&1; tail -15 $O/tests_mr.log
