#!/usr/bin/env python3
"""Phase clock shares of expand_scatter_kernel inside the real pipeline (HSK_LIB must point at a -DHSK_DIAG build)."""
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
    L.hsk_debug_diag(out, 16, 2)
    c.count_device(dp, nb, do, dl, nreads)
    L.hsk_debug_diag(out, 16, 2)
    names = ["tile claim + prologue", "window + roll + rank", "sync", "scan + reservation issue", "sync + permute", "reservation + chunk resolve",
             "sync + run stores + sync"]
    nf = max(out[10], 1)
    tot = sum(out[i] for i in range(7))
    print("flushes", out[10], "keys per flush %.0f" % (out[11] / nf), "clocks per flush %.0f" % (tot / nf))
    print("  prologue parts per flush: take prefetched + next meta %.0f, scan %.0f, tables + sync %.0f, first windows %.0f" % (out[12] / nf, out[13] / nf, out[14] / nf, out[0] / nf))
    out[0] += out[12] + out[13] + out[14]
    tot = sum(out[i] for i in range(7))
    for i, nm in enumerate(names):
        print("  %-30s %8.0f  %5.1f%%" % (nm, out[i] / nf, 100.0 * out[i] / tot))
