import sys; sys.path.insert(0, '/root/repo')
from tests import _combine_worker as W
BASE = dict(K=51, M=17, L=1, U=65535, ntasks=16, genome=1500000, read_len=150, nreads=400000, seed=77, calls=["device"], tuning="combine_min_bytes=0,scan_place=1")
print(W.run_spec(BASE))
