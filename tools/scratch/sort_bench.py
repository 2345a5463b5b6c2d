#!/usr/bin/env python3
"""Times hsk_stage_sort's device part through the profile stats (scatter launches only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hysortk_amd as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
rng = np.random.default_rng(1)
keys = rng.integers(0, 1 << 62, size=n, dtype=np.uint64) << np.uint64(2)
with H.Context(profile=True) as c:
    c.stage_sort(keys[:1000000]); c.stats()
    c.stage_sort(keys)
    st = c.stats()
    print("pad=%s launches %d avg %.3f ms  %.0f GB/s" % (os.environ.get("HSK_SORT_LDS_PAD", "0"), st["scatter_launches"], st["scatter_ms"] / st["scatter_launches"],
          st["scatter_bytes"] / st["scatter_ms"] / 1e6))
