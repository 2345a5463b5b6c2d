#!/bin/bash
# round 3: whole GPU suite, then the full default bench line with a summary
export TMPDIR=/tmp
TAG=${1:-r3full}
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/$TAG/pytest.log 2>&1
rc=$?
tail -22 gpurun_out/$TAG/pytest.log
[ $rc -ne 0 ] && exit $rc
SECONDS=0; timeout -k 10 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
rc=$?
echo "bench.py wall: ${SECONDS}s rc=$rc"; tail -3 gpurun_out/$TAG/bench.err | cut -c1-300
python3 tools/bench_summary.py gpurun_out/$TAG/bench.json
exit $rc
