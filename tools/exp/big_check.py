"""Size-independent checks of ONE call far beyond the benchmark's size (default 80 Gbp on one GPU, L = 1, U = 65535, the list left in HBM):
the count histogram adds up to the number of k-mers and to the number of entries; a second call gives the same histogram.
   python tools/exp/big_check.py [Gbp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hysortk_amd as H
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 80.0
RL = 150
G = int(gbp * 1e9) // 32
NR = G * 32 // RL
total = NR * (RL - 31 + 1)
ctx = H.Context(K=31, M=17, L=1, U=65535, ntasks=0, keep_device=True, profile=True)
dp, nb, do, dl = ctx.synth_reads(G, RL, NR, 424242)
hs = []
for it in range(2):
    t = time.perf_counter()
    r = ctx.count_device(dp, nb, do, dl, NR)
    dt = time.perf_counter() - t
    h = np.asarray(r.histo, dtype=np.uint64)
    n = int(r.info["n"]); tk = int(r.info["total_kmers"])
    s_cnt = int((h * np.arange(h.size, dtype=np.uint64)).sum(dtype=np.uint64)); s_ent = int(h.sum(dtype=np.uint64))
    print("call %d: %.1f ms, %d tasks, %d entries, total_kmers %d (expected %d), sum count*histo %d, sum histo %d" % (it, dt * 1e3, r.info["ntasks"], n, tk, total, s_cnt, s_ent), flush=True)
    assert tk == total and s_cnt == total and s_ent == n
    hs.append(h.copy()); del r
assert np.array_equal(hs[0], hs[1])
print("OK: %.0f Gbp, %.3e k-mers, genome %.2e distinct positions" % (gbp, total, G))
