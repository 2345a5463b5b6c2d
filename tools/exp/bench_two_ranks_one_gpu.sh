#!/bin/bash
# `bench.py --gpus 2` at a size where the several-rank plan is the one a real node runs (estimate, owner-side combining extraction), both ranks on
# the one GPU of the box through the stand-in transport (tests/fakerccl):   tools/exp/bench_two_ranks_one_gpu.sh [scale]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
LIB=$(python3 -c "from tests import fakerccl; print(fakerccl.build())")
HSA_ENABLE_IPC_MODE_LEGACY=0 HSK_RCCL_LIB=$LIB HSK_FORCE_DEVICE=0 HSK_FAKERCCL_TIMEOUT=120 HSK_TIMING=${HSK_TIMING:-} timeout -k 10 500 python3 bench.py --gpus 2 --scale ${1:-0.25} --steps 2 --warmup 1
