#!/bin/bash
# round 3: the two-rank tests over the stand-in transport (one GPU) + the new host-path test; logs under gpurun_out/$1
export TMPDIR=/tmp
TAG=${1:-r3a}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_host_path.py -x -q -m gpu --durations=20 > gpurun_out/$TAG/pytest.log 2>&1
rc=$?
tail -40 gpurun_out/$TAG/pytest.log
exit $rc
