"""Fuzz of the host path (diagnostic, GPU box): reads of random length distributions — fixed, almost fixed, uniform, heavy-tailed with
reads longer than a DMA slab, empty reads, reads shorter than K — counted from pinned host memory (slab ingest pipelined with scan and
placement), from pageable memory (plain copies) and, below a size limit, by the CPU oracle; the three lists must be equal.

usage: python tools/fuzz_host_path.py [first_seed] [n_seeds] [max_bases]"""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hysortk_amd as H
from hysortk_amd import synth


def variable_reads(rng, genome_len, total_bases, kind):
    g = synth.genome_codes(genome_len, int(rng.integers(1, 1 << 30)))
    if kind == "fixed":
        L = int(rng.integers(40, 400)); lens = np.full(max(1, total_bases // L), L, np.int64)
    elif kind == "almost":
        L = int(rng.integers(60, 300)); lens = np.full(max(1, total_bases // L), L, np.int64)
        for _ in range(int(rng.integers(1, 4))):
            lens[int(rng.integers(0, lens.size))] = int(rng.integers(0, 2 * L))
    elif kind == "uniform":
        hi = int(rng.integers(50, 2000)); n = max(1, 2 * total_bases // hi)
        lens = rng.integers(0, hi, n).astype(np.int64)
    else:  # heavy tail: most reads short, a few very long (longer than a slab of a small input)
        n = max(1, total_bases // 3000)
        lens = np.minimum((rng.pareto(1.1, n) * 400).astype(np.int64), genome_len - 1)
        lens[rng.integers(0, n, max(1, n // 50))] = 0
    lens = np.minimum(lens, genome_len - 1)
    start = (rng.integers(0, 1 << 62, lens.size) % (genome_len - lens)).astype(np.int64)
    pad = (lens + 3) // 4 * 4
    poff = np.concatenate([[0], np.cumsum(pad)]).astype(np.int64)
    tot = int(poff[-1])
    within = np.arange(tot, dtype=np.int64) - np.repeat(poff[:-1], pad)
    valid = within < np.repeat(lens, pad)
    gi = np.repeat(start, pad) + within
    codes = np.where(valid, g[np.minimum(gi, genome_len - 1)], 0).astype(np.uint8).reshape(-1, 4)
    packed = ((codes[:, 0] << 6) | (codes[:, 1] << 4) | (codes[:, 2] << 2) | codes[:, 3]).astype(np.uint8)
    if packed.size == 0:
        packed = np.zeros(1, np.uint8)
    return np.ascontiguousarray(packed), (poff[:-1] // 4).astype(np.uint64), lens.astype(np.uint32)


def digest(r):
    pay = b"" if r.pos is None else np.sort(r.pos.astype(np.uint64) | (r.rid.astype(np.uint64) << np.uint64(32))).tobytes()
    return hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes() + r.histo.tobytes() + pay).hexdigest()


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    max_bases = int(sys.argv[3]) if len(sys.argv) > 3 else 300_000_000
    oracle = None
    try:
        from oracle import hsk_oracle as O
        oracle = O
    except Exception as e:  # the tool still compares the three product paths
        print("oracle not available:", e)
    bad = 0
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(seed)
        kind = ("fixed", "almost", "uniform", "heavy")[seed % 4]
        K = int(rng.choice([17, 21, 31, 31, 31, 33, 51, 63, 77]))
        M = int(rng.integers(7, min(K, 26))); M = min(M, K - 1)
        EXT = int(rng.random() < 0.2)
        ntasks = int(rng.choice([1, 3, 8, 16, 40, 96, 200, 500]))
        total = int(rng.choice([2_000_000, 40_000_000, 150_000_000, max_bases]))
        total = min(total, max_bases)
        if EXT:
            total = min(total, 60_000_000)
        G = max(100_000, total // int(rng.choice([2, 8, 30])))
        Lc = int(rng.choice([1, 2, 3])); Uc = int(rng.choice([50, 255, 65535]))
        t0 = time.time()
        packed, off, lens = variable_reads(rng, G, total, kind)
        pp, po, pl = H.pinned_empty(packed.size, np.uint8), H.pinned_empty(off.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
        pp[:] = packed; po[:] = off; pl[:] = lens
        tag = f"seed {seed} {kind} K={K} M={M} EXT={EXT} ntasks={ntasks} L={Lc} U={Uc} bases={int(lens.sum())} reads={lens.size} maxlen={int(lens.max())}"
        try:
            with H.Context(K=K, M=M, L=Lc, U=Uc, EXT=EXT, ntasks=ntasks) as c:
                a = c.count((pp, po, pl)); da, na = digest(a), len(a)
                b = c.count((packed, off, lens)); db = digest(b)
                a2 = c.count((pp, po, pl)); da2 = digest(a2)
                ok = (da == db == da2)
                if oracle is not None and int(lens.sum()) <= 45_000_000 and ok:
                    want = oracle.count(packed, off, lens, k=K, m=M, L=Lc, U=Uc, ext=EXT, ntasks=ntasks, fast=True)
                    ok = np.array_equal(want.task_off, a.task_off) and np.array_equal(want.cnt, a.cnt) and np.array_equal(want.keys, a.kmers)
                    if not ok:
                        print("     differs from the oracle", len(want.cnt), na)
            print(("OK   " if ok else "FAIL ") + tag + f" entries={na} {time.time() - t0:.1f}s", flush=True)
            bad += (not ok)
        except Exception as e:
            print("EXC  " + tag + f": {e}", flush=True); bad += 1
        for y in (pp, po, pl):
            H.pinned_free(y)
    print("failures:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
