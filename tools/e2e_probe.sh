for cfg in "HSK_LAG=0" "HSK_LAG=1 HSK_STAGE2_FIRST=0"; do
  echo "== $cfg"; env $cfg HSK_TIMING=1 python tools/e2e_probe.py 2>&1 | tail -13 | cut -c1-200
done
