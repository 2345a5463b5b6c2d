#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (full BASELINE workload, 1 warmup + 1 step)
export TMPDIR=/tmp
TAG=${1:-r01}
mkdir -p gpurun_out/prof_$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/err.txt
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof_$TAG/err.txt
cat gpurun_out/prof_$TAG/bench.json | cut -c1-400
find gpurun_out/prof_$TAG -name "*stats*" | head
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cat "$f" | head -30
# keep only the summaries (the per-dispatch trace is large)
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
