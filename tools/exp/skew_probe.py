"""What a heavily repeated minimizer costs: the benchmark's reads (scaled) with a share of the reads replaced by all-A reads (one k-mer, one
minimizer bucket), counted from device memory.   python tools/exp/skew_probe.py [scale] [percent ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hysortk_amd as H

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
pcts = [float(x) for x in sys.argv[2:]] or [0.0, 0.5, 2.0]
EXT = int(os.environ.get("SKEW_EXT", "0"))
KK = int(os.environ.get("SKEW_K", "31")); PLAN = os.environ.get("SKEW_PLAN") or None; RANKS = int(os.environ.get("SKEW_RANKS", "1"))
RL = 150
G = int(312_500_000 * scale); NR = G * 32 // RL
ctx = H.Context(K=KK, M=17, L=15, U=40, EXT=EXT, ntasks=0 if RANKS == 1 else 40 * RANKS, profile=True, keep_device=True, plan=PLAN)
dp, nb, do, dl = ctx.synth_reads(G, RL, NR, 7)
packed = H.pinned_empty(nb, np.uint8); ctx.d2h_into(packed, dp, nb)
off = H.pinned_empty(NR, np.uint64); lens = H.pinned_empty(NR, np.uint32)
ctx.d2h_into(off, do, NR * 8); ctx.d2h_into(lens, dl, NR * 4)
ctx.synth_free(dp, do, dl)
bpr = (RL + 3) // 4
view = packed.reshape(NR, bpr)
clean = np.array(view, copy=True)
rng = np.random.default_rng(3)
for pct in pcts:
    view[:] = clean
    n = int(NR * pct / 100)
    if n:
        view[rng.choice(NR, n, replace=False)] = 0          # all-A reads
    for it in range(3):
        ctx.stats(reset=True)
        t = time.perf_counter()
        if RANKS == 1:
            r = ctx.count((packed, off, lens))
            info = dict(r.info)
        else:                                  # the several-rank path on this GPU: the reads dealt to RANKS virtual ranks
            per = NR // RANKS
            parts = [(packed[i * per * bpr:(i + 1) * per * bpr], off[:per], lens[:per]) for i in range(RANKS)]
            r, owner = ctx.count_loopback(parts)
            info = dict(r[0].info)
        dt = time.perf_counter() - t
        st = ctx.stats(reset=True)
        del r
    print("all-A reads %.1f %%: %.1f ms host to host, combine launches %d, phases %s" % (pct, dt * 1e3, st["combine_launches"], {k[3:]: round(v, 1) for k, v in info.items() if k.startswith("ms_")}), flush=True)
