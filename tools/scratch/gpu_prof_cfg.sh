#!/bin/bash
# rocprofv3 kernel stats of one bench configuration: tools/gpu_prof_cfg.sh <tag> <bench args...>
export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e "$@" > $OUT/bench.json 2> $OUT/bench.err < /dev/null
cd $GRAFT_REPO_ROOT
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then head -14 "$F" | cut -c1-160; else echo "no kernel stats"; tail -5 $OUT/bench.err; fi
find $OUT -name "*kernel_trace.csv" -delete
