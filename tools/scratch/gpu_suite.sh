#!/bin/bash
# whole GPU suite (log under gpurun_out/) followed by one bench line without the cpu leg
export TMPDIR=/tmp
TAG=${1:-suite}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/$TAG/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/$TAG/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rc=$?
tail -3 gpurun_out/$TAG/bench.err; cut -c1-1500 gpurun_out/$TAG/bench.json
exit $rc
