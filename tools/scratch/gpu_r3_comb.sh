#!/bin/bash
# combining extraction: parity subset (forced on for every input size), then kernel statistics of the bench
export TMPDIR=/tmp
tag=${1:-comb}
mkdir -p gpurun_out/$tag
HSK_COMBINE_MIN_BYTES=0 timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${2:-synth_vs_oracle or random_configurations or golden or fused_scatter_sweep or full_size or edge or s_ecoli or long_records}" > gpurun_out/$tag/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/$tag/pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_r3_combprof.sh $tag/prof
