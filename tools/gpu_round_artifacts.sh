#!/bin/bash
# end-of-round evidence: full bench line (with cpu_baseline), rocprofv3 kernel stats of the same command,
# PMC traffic (FETCH_SIZE / WRITE_SIZE passes) at full scale
export TMPDIR=/tmp
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$set -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/pmc_$set.json 2> $OUT/pmc_$set.err
done
cd $GRAFT_REPO_ROOT
python3 tools/traffic_from_pmc.py $OUT $OUT/traffic.json
mkdir -p profiles && cp $OUT/traffic.json profiles/traffic.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cat $(find $OUT/stats -name "*kernel_stats.csv" | head -1) | cut -c1-160 | head -14
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
