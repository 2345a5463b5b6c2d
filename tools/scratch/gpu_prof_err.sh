#!/bin/bash
# rocprofv3 kernel-trace summary of the bench workload with substitution errors (scale and rate from the arguments)
export TMPDIR=/tmp
SCALE=${1:-0.4}; ER=${2:-0.01}; TAG=${3:-proferr}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --scale $SCALE --error-rate $ER > $OUT/bench.json 2> $OUT/err.txt
cd $GRAFT_REPO_ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -22 "$f" | cut -c1-170
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print('value %.2f G  %.1f ms' % (d['value']/1e9, d['ms_per_step']), d['path_stats'])"
