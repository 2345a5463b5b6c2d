#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/combtest
timeout -k 10 1000 python -m pytest tests/test_gpu_combine.py -x -q -m gpu > gpurun_out/combtest/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/combtest/pytest.log
exit $rc
