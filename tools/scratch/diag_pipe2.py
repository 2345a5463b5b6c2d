#!/usr/bin/env python3
"""Phase clock shares of onesweep_tile inside the real pipeline (HSK_LIB must point at a -DHSK_DIAG build)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hysortk_amd as H
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
G = int(312_500_000 * scale); nreads = G * 32 // 150
with H.Context(K=31, M=17, L=15, U=40, keep_device=True) as c:
    L = c.lib
    L.hsk_debug_diag.argtypes = [C.c_void_p, C.c_int, C.c_int]
    out = (C.c_ulonglong * 32)()
    dp, nb, do, dl = c.synth_reads(G, 150, nreads, 1)
    c.count_device(dp, nb, do, dl, nreads)
    L.hsk_debug_diag(out, 32, 1)
    c.count_device(dp, nb, do, dl, nreads)
    L.hsk_debug_diag(out, 32, 1)
    names = ["ticket+zero+sync", "load keys (vmcnt0)", "rank", "sync", "scan+publish+permute", "lookback", "sync", "scatter issue", "drain stores"]
    tot = sum(out[i] for i in range(9))
    print("tiles", out[16], "avg cycles per tile", tot / max(out[16], 1))
    for i, nm in enumerate(names):
        print("  %-24s %8.0f cyc  %5.1f%%" % (nm, out[i] / max(out[16], 1), 100.0 * out[i] / tot))
    print("  look-back per tile (digit 0): window steps %.2f, not-ready retries %.2f, depth %.1f tiles" % (out[10] / out[16], out[11] / out[16], out[12] / out[16]))
    sn = ["stage + index probe", "hash", "window min + validity", "boundaries + compaction", "dense (task, run, record)"]
    st = sum(out[20 + i] for i in range(5))
    if out[28]:
        print("scan_kernel tiles", out[28], "avg clocks per tile", st / out[28])
        for i, nm in enumerate(sn):
            print("  %-28s %8.0f  %5.1f%%" % (nm, out[20 + i] / out[28], 100.0 * out[20 + i] / st))
