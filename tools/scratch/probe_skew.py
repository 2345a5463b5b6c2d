#!/usr/bin/env python3
"""Throughput on extreme multiplicities (a tiny genome at very high coverage: few distinct k-mers, thousands of copies each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hysortk_amd as H

def run(tag, G, read_len, cov, K=31, ext=0):
    nreads = int(G * cov // read_len)
    with H.Context(K=K, M=17, L=2, U=65535, EXT=ext, keep_device=True) as c:
        dp, nb, do, dl = c.synth_reads(G, read_len, nreads, 3)
        best = None
        for it in range(3):
            t0 = time.time(); r = c.count_device(dp, nb, do, dl, nreads); dt = time.time() - t0
            best = dt if best is None or dt < best else best
        st = c.stats()
        print("%-40s %6.2f G k-mers/s  %7.1f ms  redone %d retried %d heavy %d" % (tag, nreads * max(read_len - K + 1, 0) / best / 1e9, best * 1e3, st["redone_tasks"], st["agg_retried_tasks"], st["heavy_tasks"]), flush=True)
        c.synth_free(dp, do, dl)

run("1 Mbp genome x 3000 coverage", 1_000_000, 150, 3000)
run("100 kbp genome x 30000 coverage", 100_000, 150, 30000)
run("10 kbp genome x 300000 coverage", 10_000, 150, 300000)
run("1 Mbp genome x 3000 coverage, EXT", 1_000_000, 150, 3000, ext=1)
run("1 Mbp genome x 3000 coverage, K=51", 1_000_000, 150, 3000, K=51)
