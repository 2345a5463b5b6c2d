"""Fuzz of the skew machinery (diagnostic, GPU box): reads of a random genome with random shares replaced by homopolymer reads (A, C, G, T) and
dinucleotide reads, random K / EXTENSION / U / task counts, counted from pinned host memory with the library's defaults and with everything that
treats skew switched off (drop_certain=0, agg_large=0, combine=0): lists, histograms and (EXTENSION) every k-mer's payload multiset must be equal.
usage: python tools/fuzz_skew.py [first_seed] [n_seeds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hysortk_amd as H
from tests._combine_worker import digest

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
fails = 0
for seed in range(first, first + nseeds):
    rng = np.random.default_rng(seed)
    K = int(rng.choice([21, 31, 41, 51])); EXT = int(rng.random() < 0.3); U = int(rng.choice([40, 200, 65535])); L = int(rng.choice([1, 2, 15]))
    RL = int(rng.choice([100, 150, 250])); n = int(rng.integers(1_000_000, 2_200_000)); G = int(n * RL / rng.choice([8, 32]))
    ntasks = int(rng.choice([0, 8, 24, 40]))
    res = []
    shares = {b: float(rng.choice([0, 0, 0.5, 2, 6])) for b in "ACGT"}; di = float(rng.choice([0, 0, 1, 4]))
    for tun in (None, "drop_certain=0,agg_large=0,combine=0"):
        ctx = H.Context(K=K, M=17, L=L, U=U, EXT=EXT, ntasks=ntasks, profile=True, tuning=tun)
        dp, nb, do, dl = ctx.synth_reads(G, RL, n, seed)
        packed = H.pinned_empty(nb, np.uint8); off = H.pinned_empty(n, np.uint64); lens = H.pinned_empty(n, np.uint32)
        ctx.d2h_into(packed, dp, nb); ctx.d2h_into(off, do, n * 8); ctx.d2h_into(lens, dl, n * 4)
        ctx.synth_free(dp, do, dl)
        r2 = np.random.default_rng(seed + 1000)
        view = packed.reshape(n, (RL + 3) // 4)
        order = r2.permutation(n); at = 0
        for b, code in zip("ACGT", (0x00, 0x55, 0xAA, 0xFF)):
            m = int(n * shares[b] / 100); view[order[at:at + m]] = code; at += m
        m = int(n * di / 100); view[order[at:at + m]] = 0x11                      # ACAC...
        t = time.perf_counter(); r = ctx.count((packed, off, lens)); dt = time.perf_counter() - t
        st = ctx.stats()
        res.append((digest(r), len(r), int(r.info["total_kmers"]), int(st["dropped_kmers"]), dt))
        del r
        for x in (packed, off, lens): H.pinned_free(x)
        ctx.close()
    ok = res[0][:3] == res[1][:3]
    fails += not ok
    print("%s seed %d K=%d EXT=%d L=%d U=%d RL=%d reads=%d cov=%.0f ntasks=%d homopolymers %s di %.0f%%: entries %d, dropped %d, %.0f ms against %.0f ms" %
          ("OK  " if ok else "FAIL", seed, K, EXT, L, U, RL, n, n * RL / G, ntasks, {k: v for k, v in shares.items() if v}, di, res[0][1], res[0][3], res[0][4] * 1e3, res[1][4] * 1e3), flush=True)
print("failures:", fails)
