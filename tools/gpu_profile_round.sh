#!/bin/bash
# THE script behind profiles/<tag>_*: everything the bench line's roofline block quotes, from the build that is in the tree.
#   tools/gpu_profile_round.sh r04            (on the GPU box; copy gpurun_out/<tag>/profiles/* into profiles/ afterwards, or run
#                                              tools/collect_profiles.sh <tag> here)
# 1. rocprofv3 --kernel-trace --stats of the bench command (3 steps + 1 warm-up)            -> <tag>_kernel_stats.csv
# 2. PMC passes, one counter set each, kernel trace only (1 step + 1 warm-up, full scale):   -> <tag>_pmc.json (+ pmc.json, the one bench.py reads)
#      FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES | SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
#                                                                                               | SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
# 3. tools/exp/mulrate                                                                        -> <tag>_valu_rates.txt; tools/valu_floor.py -> valu_mix.json
# 4. the full bench line (reads the pmc.json of step 2)                                       -> <tag>_bench.json
export TMPDIR=/tmp
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
P=$OUT/profiles
mkdir -p $OUT $P
CMD="python3 $R/bench.py --no-cpu --no-e2e --no-variants"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $CMD --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $P/${TAG}_kernel_stats.csv
cp $OUT/bench_under_rocprof.json $P/${TAG}_bench_under_rocprof.json
echo "[profile] kernel stats done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc/p$i -o p$i -- $CMD --steps 1 --warmup 1 > $OUT/pmc_p$i.json 2> $OUT/pmc_p$i.err
  echo "[profile] pmc pass $i ($set) done"
done
cd $R
python3 tools/pmc_to_json.py $OUT/pmc $P/${TAG}_pmc.json 2 $OUT/pmc_p1.json > $OUT/pmc_summary.txt
cp $P/${TAG}_pmc.json $R/profiles/pmc.json
cat $OUT/pmc_summary.txt
[ -x tools/exp/mulrate ] && ./tools/exp/mulrate > $P/${TAG}_valu_rates.txt 2>&1
# the kernels' static mix priced with THIS run's rates, before the bench line that quotes it (30 s of hipcc -S; round 4 computed it beforehand
# from the previous run's rates: 0.5 % apart)
python3 tools/valu_floor.py $P/${TAG}_valu_rates.txt profiles/valu_mix.json > $OUT/valu_mix.txt 2>&1 && cp profiles/valu_mix.json $P/valu_mix.json
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/bench.json $P/${TAG}_bench.json
python3 tools/bench_summary.py $OUT/bench.json
head -14 $P/${TAG}_kernel_stats.csv | cut -c1-160
find $OUT -name "*kernel_trace.csv" -size +4M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
