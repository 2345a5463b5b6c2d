#!/usr/bin/env python3
"""K=77 through the batch path on plain random reads: stats and comparison with the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hysortk_amd as H
import util
O = util.oracle() if hasattr(util, "oracle") else None
rng = np.random.default_rng(1)
g = "".join(rng.choice(list("ACGT"), 50000))
reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 20000)]
dna = H.DnaBuffer.from_sequences(reads)
for K in (77,):
    with H.Context(K=K, M=17, L=1, U=65535, ntasks=16) as c:
        res = c.count(dna)
        st = c.stats()
    print(K, {k: st[k] for k in ("fused_tasks", "redone_tasks", "agg_retried_tasks")}, len(res))
    import ctypes as C
    L = H._lib.load() if hasattr(H, "_lib") else None
    out = (C.c_ulonglong * 16)()
    L.hsk_debug_diag.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.hsk_debug_diag(out, 16, 4)
    print("diag: no-slot", out[0], "timeouts", out[1], "bin iterations", out[2], "records", out[3])
