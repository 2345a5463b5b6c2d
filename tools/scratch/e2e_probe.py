"""e2e_host leg alone with HSK_TIMING marks (diagnostic)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hysortk_amd as H
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
g = int(bench.GENOME_PER_GPU * scale)
r = bench.e2e_host_leg(H, 31, 0, 0, 0, g, g * bench.COVERAGE // bench.READ_LEN, 20251003, 2)
print(json.dumps(r))
