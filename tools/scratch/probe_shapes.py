#!/usr/bin/env python3
"""Throughput over input shapes the benchmark does not cover (coverage, read length): looking for cliffs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hysortk_amd as H

def run(tag, G, read_len, cov, K=31, ext=0, err=0.0):
    nreads = int(G * cov // read_len)
    with H.Context(K=K, M=17, L=2, U=65535, EXT=ext, keep_device=True) as c:
        dp, nb, do, dl = c.synth_reads(G, read_len, nreads, 3, error_rate=err)
        best = None
        for it in range(3):
            t0 = time.time()
            r = c.count_device(dp, nb, do, dl, nreads)
            dt = time.time() - t0
            best = dt if best is None or dt < best else best
            n = r.total_kmers if hasattr(r, "total_kmers") else nreads * max(read_len - K + 1, 0)
        st = c.stats()
        print("%-34s %6.2f G k-mers/s  %7.1f ms  entries %d  redone %d retried %d" % (tag, nreads * max(read_len - K + 1, 0) / best / 1e9, best * 1e3, len(r), st["redone_tasks"], st["agg_retried_tasks"]), flush=True)
        c.synth_free(dp, do, dl)

G = 200_000_000
run("cov 32 x 150 bp (reference shape)", G, 150, 32)
run("cov 5 x 150 bp", G * 4, 150, 5)
run("cov 2 x 150 bp", G * 8, 150, 2)
run("cov 100 x 150 bp", G // 4, 150, 100)
run("cov 32 x 50 bp", G, 50, 32)
run("cov 32 x 10 kbp", G, 10000, 32)
run("cov 32 x 1 Mbp", G, 1000000, 32)
run("cov 5 x 150 bp, EXT", G * 2, 150, 5, ext=1)
run("cov 5 x 150 bp, K=51", G * 2, 150, 5, K=51)
