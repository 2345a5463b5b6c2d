#!/bin/bash
# gpurun merges only gpurun_out/ back: copy the round's evidence from gpurun_out/<tag>/ into profiles/ (tracked)
TAG=${1:-r03}
O=gpurun_out/$TAG
cp $O/traffic.json profiles/traffic.json
cp $O/traffic.json profiles/${TAG}_traffic.json
[ -f $O/traffic_place_bytes.json ] && cp $O/traffic_place_bytes.json profiles/${TAG}_traffic_place_bytes.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) profiles/${TAG}_kernel_stats.csv
cp $O/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
cp $O/bench.json profiles/${TAG}_bench.json
ls -la profiles | grep $TAG
