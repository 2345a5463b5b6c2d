#!/bin/bash
# end-of-round evidence: full bench line (with e2e_host and cpu_baseline), rocprofv3 kernel stats of the same command,
# PMC traffic (FETCH_SIZE / WRITE_SIZE passes) at full scale; summaries copied to profiles/ under the round's tag
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$set -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --no-variants > $OUT/pmc_$set.json 2> $OUT/pmc_$set.err
done
cd $GRAFT_REPO_ROOT
python3 tools/traffic_from_pmc.py $OUT $OUT/traffic.json
mkdir -p profiles && cp $OUT/traffic.json profiles/traffic.json && cp $OUT/traffic.json profiles/${TAG}_traffic.json
# the same two passes on the instance path with the byte-store placement forced on (one GPU: off by default): what the extraction fetches then
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  HSK_COMBINE=0 HSK_PLACE_BYTES=1 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/bytes/pmc_$set -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --no-variants > $OUT/bytes_pmc_$set.json 2> $OUT/bytes_pmc_$set.err
done
cd $GRAFT_REPO_ROOT
python3 tools/traffic_from_pmc.py $OUT/bytes $OUT/traffic_place_bytes.json > /dev/null && cp $OUT/traffic_place_bytes.json profiles/${TAG}_traffic_place_bytes.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-variants > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cd $GRAFT_REPO_ROOT
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) profiles/${TAG}_kernel_stats.csv
cp $OUT/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/bench.json profiles/${TAG}_bench.json
python3 tools/bench_summary.py $OUT/bench.json
head -14 profiles/${TAG}_kernel_stats.csv | cut -c1-160
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
