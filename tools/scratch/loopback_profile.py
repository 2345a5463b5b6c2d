#!/usr/bin/env python3
"""Virtual-rank run (hsk_count_loopback) at a size where kernel times are meaningful: shows what the multi-GPU-only
kernels (byte packing, multi-segment expand) cost next to the rest.  Run under rocprofv3 --kernel-trace --stats."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hysortk_amd as H

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nreads = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
c = H.Context(K=31, M=17, L=15, U=40, ntasks=16 * R)
parts = []
for r in range(R):
    dp, nb, do, dl = c.synth_reads(nreads * R * 150 // 32, 150, nreads, 7, first_read=r * nreads)
    packed = c.d2h(dp, nb)
    c.synth_free(dp, do, dl)
    nbr = 38
    off = np.arange(nreads, dtype=np.uint64) * np.uint64(nbr)
    lens = np.full(nreads, 150, dtype=np.uint32)
    parts.append(H.DnaBuffer.from_arrays(packed, off, lens))
for it in range(2):
    t0 = time.time()
    res, owner = c.count_loopback(parts)
    print("loopback R=%d: %.1f ms, entries %d" % (R, (time.time() - t0) * 1e3, sum(len(k) for k in res)))
