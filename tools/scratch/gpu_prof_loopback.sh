#!/bin/bash
# rocprofv3 kernel stats of a virtual-rank run (hsk_count_loopback): tools/gpu_prof_loopback.sh <tag> <ranks> <reads per rank>
export TMPDIR=/tmp
TAG=$1; R=${2:-2}; N=${3:-8000000}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o prof -- python3 $GRAFT_REPO_ROOT/tools/loopback_profile.py $R $N > $OUT/run.log 2> $OUT/run.err < /dev/null
cd $GRAFT_REPO_ROOT
cat $OUT/run.log
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then head -16 "$F" | cut -c1-170; else echo "no kernel stats"; tail -5 $OUT/run.err; fi
find $OUT -name "*kernel_trace.csv" -delete
