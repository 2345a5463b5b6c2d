#!/bin/bash
O=gpurun_out/${1:-r4e}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_combine.py tests/test_gpu_rccl.py tests/test_gpu_host_path.py -x -q -m gpu > $O/tests_a.log 2>&1; tail -12 $O/tests_a.log
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "switch or plans or fused or adapt or wide or ext_" > $O/tests_b.log 2>&1; tail -8 $O/tests_b.log
