#!/usr/bin/env python3
"""Phase clock shares of agg_finish_kernel inside the real pipeline (HSK_LIB must point at a -DHSK_DIAG build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hysortk_amd as H
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
G = int(312_500_000 * scale); nreads = G * 32 // 150
with H.Context(K=31, M=17, L=15, U=40, keep_device=True) as c:
    L = c.lib
    L.hsk_debug_diag.argtypes = [C.c_void_p, C.c_int, C.c_int]
    out = (C.c_ulonglong * 16)()
    dp, nb, do, dl = c.synth_reads(G, 150, nreads, 1)
    c.count_device(dp, nb, do, dl, nreads)
    L.hsk_debug_diag(out, 16, 4)
    c.count_device(dp, nb, do, dl, nreads)
    L.hsk_debug_diag(out, 16, 4)
    names = ["launch -> bounds known", "table cleared", "records loaded + counted", "distinct keys compacted", "distinct keys ordered", "filtered + written"]
    nwg = max(out[8], 1); tot = sum(out[i] for i in range(6))
    print("workgroups %d, records per bin %.0f, distinct per bin %.1f, clocks per workgroup %.0f (100 MHz ticks?)" % (out[8], out[9] / nwg, out[10] / nwg, tot / nwg))
    for i, nm in enumerate(names):
        print("  %-28s %8.0f  %5.1f%%" % (nm, out[i] / nwg, 100.0 * out[i] / tot))
