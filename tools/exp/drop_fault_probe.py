"""one case of the fallback + certain drops fault hunt: python tools/exp/drop_fault_probe.py '<tuning_extra>' [call]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import _combine_worker as W
BASE = dict(K=31, M=17, L=2, U=40, ntasks=0, genome=8000000, read_len=150, nreads=1800000, seed=91, poly_a_pct=5.0)
spec = dict(BASE, calls=[sys.argv[2] if len(sys.argv) > 2 else "pinned"], tuning=sys.argv[1])
print(json.dumps(W.run_spec(spec)))
