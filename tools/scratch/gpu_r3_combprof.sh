#!/bin/bash
export TMPDIR=/tmp
TAG=${1:-combprof}
mkdir -p gpurun_out/$TAG
HSK_TIMING=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-variants > gpurun_out/$TAG/b0.json 2> gpurun_out/$TAG/b0.err
grep "combining" gpurun_out/$TAG/b0.err | head -12
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-variants > $GRAFT_REPO_ROOT/gpurun_out/$TAG/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/$TAG/err.txt
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -24 "$f" | cut -c1-200
find gpurun_out/$TAG -name "*kernel_trace.csv" -size +20M -delete
