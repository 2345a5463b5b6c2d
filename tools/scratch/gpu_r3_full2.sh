#!/bin/bash
# full GPU suite twice (defaults; the combining extraction forced on for every input size), then the bench line
export TMPDIR=/tmp
tag=${1:-full2}
mkdir -p gpurun_out/$tag
if [ -z "$SKIP_DEFAULT" ]; then timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/pytest_default.log 2>&1; rc=$?; else rc=0; fi
tail -4 gpurun_out/$tag/pytest_default.log
[ $rc -ne 0 ] && exit $rc
HSK_COMBINE_MIN_BYTES=0 timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_combine.py > gpurun_out/$tag/pytest_forced.log 2>&1; rc=$?
tail -12 gpurun_out/$tag/pytest_forced.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err && python tools/bench_summary.py gpurun_out/$tag/bench.json
