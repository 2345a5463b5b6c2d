#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) on a reduced workload; aggregates per kernel.
export TMPDIR=/tmp
TAG=${1:-pmc}
SCALE=${2:-0.1}
EXTRA=${3:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --scale $SCALE $EXTRA > $OUT/p$i.json 2> $OUT/p$i.err
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_agg.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +8M -delete
