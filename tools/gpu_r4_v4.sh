#!/bin/bash
O=gpurun_out/${1:-r4e}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_combine.py -x -q -m gpu > $O/tests_combine.log 2>&1; tail -12 $O/tests_combine.log
timeout -k 10 600 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu -k "owner_side or golden or groups" > $O/tests_rccl.log 2>&1; tail -12 $O/tests_rccl.log
timeout -k 10 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $O/tests_mr.log 2>&1; tail -12 $O/tests_mr.log
