python -m pytest tests/test_gpu_host_path.py -x -q -m gpu 2>&1 | tail -2
for cfg in "HSK_DERIVE_OFFSETS=1" "HSK_DERIVE_OFFSETS=0"; do
  echo "== $cfg"; env $cfg HSK_TIMING=1 python tools/e2e_probe.py 2>&1 | tail -13 | cut -c1-200
done
