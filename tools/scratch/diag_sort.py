#!/usr/bin/env python3
"""Diagnostic: phase clock shares of the onesweep kernel (needs hysortk_amd/libhsk_diag.so, built with -DHSK_DIAG)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hysortk_amd import _lib
_lib.lib_path = lambda: os.path.join(ROOT, "hysortk_amd", os.environ.get("HSK_DIAG_LIB", "libhsk_diag.so"))
from hysortk_amd import build as b
b.needs_build = lambda: False
import hysortk_amd as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 62, size=n, dtype=np.uint64) << np.uint64(2)
with H.Context() as c:
    L = c.lib
    L.hsk_debug_diag.argtypes = [C.c_void_p, C.c_int, C.c_int]
    out = (C.c_ulonglong * 32)()
    c.stage_sort(keys[:1000000])
    L.hsk_debug_diag(out, 32, 1)
    c.stage_sort(keys)
    L.hsk_debug_diag(out, 32, 1)
    names = ["ticket+zero+sync", "load keys (vmcnt0)", "rank", "sync", "scan+publish+permute", "lookback", "sync", "scatter issue", "drain stores"]
    tot = sum(out[i] for i in range(9))
    print("blocks", out[16], "avg cycles per tile", tot / max(out[16], 1))
    for i, nm in enumerate(names):
        print("  %-24s %8.0f cyc  %5.1f%%" % (nm, out[i] / max(out[16], 1), 100.0 * out[i] / tot))
    print("  look-back per tile (digit 0): window steps %.2f, not-ready retries %.2f, depth %.1f tiles" % (out[10] / out[16], out[11] / out[16], out[12] / out[16]))
