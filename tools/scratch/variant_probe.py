"""One variant leg of bench.py alone (HSK_TIMING=1 for host marks): python tools/variant_probe.py uniform|k51|ext|no_aggregation|full_sort [scale]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, hysortk_amd as H
name = sys.argv[1]; scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
G = int(bench.GENOME_PER_GPU * scale); NR = G * bench.COVERAGE // bench.READ_LEN
cfg = {"k51": (51, 0, None, 15, 40, G, NR, 0.0), "ext": (31, 1, None, 15, 40, G, NR, 0.0), "no_aggregation": (31, 0, "no_aggregation", 15, 40, G, NR, 0.0),
       "full_sort": (31, 0, "full_sort", 15, 40, G, NR, 0.0), "uniform": (31, 0, None, 1, 65535, G // 2, NR // 2, 0.75)}[name]
r = bench.run_variant(H, 0, name, "", cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], cfg[6], 20251003, cfg[7], 2, None)
print(json.dumps({k: r[k] for k in ("name", "value", "ms_per_step", "device_ms_total", "entries", "phases_ms", "scatter_pass")}))
