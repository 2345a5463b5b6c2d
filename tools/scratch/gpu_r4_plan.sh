#!/bin/bash
# round 4: the in-call plan estimate -- its tests, what it costs (HSK_TIMING marks), and the first-call leg of the bench
O=gpurun_out/${1:-r4c}; mkdir -p $O
python -m pytest tests/test_gpu_combine.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for er in 0 0.003 0.01 0.75; do
  HSK_TIMING=1 python bench.py --steps 2 --warmup 0 --no-cpu --no-variants --no-e2e --error-rate $er > $O/bench_er$er.json 2> $O/bench_er$er.err
  grep "plan estimate" $O/bench_er$er.err | head -2
done
python bench.py --steps 3 --warmup 1 --no-cpu --no-e2e > $O/bench_variants.json 2> $O/bench_variants.err
python tools/bench_summary.py $O/bench_variants.json
./tools/exp/mulrate > $O/valu_rates.txt 2>&1; tail -3 $O/valu_rates.txt | cut -c1-200
