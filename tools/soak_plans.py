"""One context, inputs that switch the plan back and forth (diagnostic, GPU box): error-free deep reads take the combining extraction,
reads with 1 % errors make the context leave it for eight calls, then it looks again (and waits twice as long after every look that finds the same).  Every call of the same input must give the
same list; the pools must not grow without bound."""
import os, sys, hashlib, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hysortk_amd as H

G, RL = 156250000, 150
NR = G * 32 // RL
ctx = H.Context(K=31, M=17, L=2, U=200, ntasks=0, profile=True, keep_device=True)
clean = ctx.synth_reads(G, RL, NR, 11)
noisy = ctx.synth_reads(G, RL, NR, 12, error_rate=0.01)
seen = {}
plan = ["clean"] * 3 + ["noisy"] * 30 + ["clean"] * 20 + ["noisy"] * 2 + ["clean"] * 2
t0 = time.time()
for i, what in enumerate(plan):
    dp, nb, do, dl = clean if what == "clean" else noisy
    ctx.stats(reset=True)
    t = time.perf_counter()
    r = ctx.count_device(dp, nb, do, dl, NR)
    dt = time.perf_counter() - t
    st = ctx.stats(reset=True)
    d = (int(r.info["n"]), int(r.info["total_kmers"]))
    del r
    assert seen.setdefault(what, d) == d, (what, d, seen[what])
    print("%2d %-5s %6.1f ms  entries %d  %s" % (i, what, dt * 1e3, d[0], "combining extraction (%.1f k-mers per pair)" % (st["combine_kmers"] / max(st["combine_pairs"], 1)) if st["combine_pairs"] else "instance path"), flush=True)
print("ok %.1f s" % (time.time() - t0))
