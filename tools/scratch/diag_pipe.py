#!/usr/bin/env python3
"""Diagnostic: onesweep phase shares inside the whole pipeline (diag build), single-task vs XCD-batched."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hysortk_amd import _lib
_lib.lib_path = lambda: os.path.join(ROOT, "hysortk_amd", "libhsk_diag.so")
from hysortk_amd import build as b
b.needs_build = lambda: False
import hysortk_amd as H
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
G = int(312_500_000 * scale); NR = G * 32 // 150
with H.Context(K=31, M=17, L=15, U=40, ntasks=8, profile=True, keep_device=True) as c:
    L = c.lib
    L.hsk_debug_diag.argtypes = [C.c_void_p, C.c_int, C.c_int]
    out = (C.c_ulonglong * 32)()
    dp, nb, do, dl = c.synth_reads(G, 150, NR, 5)
    c.count_device(dp, nb, do, dl, NR)
    L.hsk_debug_diag(out, 32, 1); c.stats()
    info = c.count_device(dp, nb, do, dl, NR).info
    st = c.stats()
    L.hsk_debug_diag(out, 32, 1)
    names = ["ticket+zero+sync", "load keys (vmcnt0)", "rank", "sync", "scan+publish+permute", "lookback", "sync", "scatter issue", "drain stores"]
    tot = sum(out[i] for i in range(9))
    print("XCD_BATCH=%s sort %.1f ms; scatter launches %d avg %.3f ms %.0f GB/s" % (os.environ.get("HSK_XCD_BATCH", "1"), info["ms_sort"], st["scatter_launches"],
          st["scatter_ms"] / max(st["scatter_launches"], 1), st["scatter_bytes"] / max(st["scatter_ms"], 1e-9) / 1e6))
    print("tiles", out[16], "avg units per tile", tot / max(out[16], 1))
    for i, nm in enumerate(names):
        print("  %-24s %8.0f  %5.1f%%" % (nm, out[i] / max(out[16], 1), 100.0 * out[i] / tot))
    print("  look-back per tile (digit 0): window steps %.2f, not-ready retries %.2f, depth %.1f tiles" % (out[10] / out[16], out[11] / out[16], out[12] / out[16]))
