#!/bin/bash
# scan_kernel with and without its own item placement: instruction counts and busy cycles (PMC, 1/5 scale)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4j}; mkdir -p $O
cd /tmp
for mode in 1 0; do
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    HSK_TUNING=scan_place=$mode rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/m$mode/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --no-variants --scale 0.2 > $O/m${mode}_p$i.json 2> $O/m${mode}_p$i.err
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_agg.py $O/m$mode > $O/summary_m$mode.txt
  echo "== scan_place=$mode"; grep -A18 "^scan_kernel" $O/summary_m$mode.txt | head -20
done
find $O -name "*.csv" -size +4M -delete
