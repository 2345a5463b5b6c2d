#!/bin/bash
# first contact with the GPU: everything under short timeouts so a hang cannot eat the box
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -20
echo "== sort stage"; timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stage_sort or stage_count" 2>&1 | tail -15
echo "== rest"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -25
