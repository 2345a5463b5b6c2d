"""Fuzz of the skew machinery (diagnostic, GPU box): reads of a random genome with random shares replaced by homopolymer reads (A, C, G, T) and
dinucleotide reads, random K / EXTENSION / U / task counts, counted from pinned host memory with the library's defaults and with everything that
treats skew switched off (drop_certain=0, agg_large=0, combine=0): lists, histograms and (EXTENSION) every k-mer's payload multiset must be equal.
Some seeds split the reads over 2 - 3 virtual ranks (hsk_count_loopback: the ranks share one verdict on the certain drops; lists compared task by
task, whoever owns the task), some force a record capacity the scan overflows (parse_rec_cap=200: the general parse kernels, same mask).
usage: python tools/fuzz_skew.py [first_seed] [n_seeds] [ranks: 0 = random]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hysortk_amd as H
from tests._combine_worker import digest

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
force_r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fails = 0


def loop_digest(res, owner):
    """digest over the tasks in task order, each from its owner's list (+ the payload multiset of every kept k-mer with EXTENSION)"""
    import hashlib
    h = hashlib.sha256(); ent = 0; cs = {}
    for t in range(len(owner)):
        o = int(owner[t]); kl = res[o]
        a, b = int(kl.task_off[t]), int(kl.task_off[t + 1])
        h.update(kl.kmers[a:b].tobytes()); h.update(kl.cnt[a:b].tobytes()); ent += b - a
        if kl.pos is not None and b > a:
            if o not in cs:
                x = kl.pos.astype(np.uint64) | (kl.rid.astype(np.uint64) << np.uint64(32))
                with np.errstate(over="ignore"):
                    cs[o] = np.concatenate((np.zeros(1, np.uint64), np.cumsum(x * np.uint64(0x9E3779B97F4A7C15), dtype=np.uint64)))
            po = kl.payload_off[a:b].astype(np.int64)
            with np.errstate(over="ignore"):
                h.update((cs[o][po + kl.cnt[a:b].astype(np.int64)] - cs[o][po]).tobytes())
    return h.hexdigest(), ent


for seed in range(first, first + nseeds):
    rng = np.random.default_rng(seed)
    K = int(rng.choice([21, 31, 41, 51])); EXT = int(rng.random() < 0.3); U = int(rng.choice([40, 200, 65535])); L = int(rng.choice([1, 2, 15]))
    RL = int(rng.choice([100, 150, 250])); n = int(rng.integers(1_000_000, 2_200_000)); G = int(n * RL / rng.choice([8, 32]))
    ntasks = int(rng.choice([0, 8, 24, 40]))
    R = force_r if force_r else int(rng.choice([1, 1, 2, 3])); cap = int(rng.choice([0, 0, 0, 200]))
    res = []
    shares = {b: float(rng.choice([0, 0, 0.5, 2, 6])) for b in "ACGT"}; di = float(rng.choice([0, 0, 1, 4]))
    for tun in ("parse_rec_cap=%d" % cap if cap else None, "drop_certain=0,agg_large=0,combine=0"):
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
        if R == 1:
            t = time.perf_counter(); r = ctx.count((packed, off, lens)); dt = time.perf_counter() - t
            st = ctx.stats()
            res.append((digest(r), len(r), int(r.info["total_kmers"]), int(st["dropped_kmers"]), dt))
        else:
            per = n // R; nbr = (RL + 3) // 4
            parts = [(packed[q * per * nbr:(q + 1) * per * nbr], off[:per], lens[:per]) for q in range(R)]
            t = time.perf_counter(); r, owner = ctx.count_loopback(parts); dt = time.perf_counter() - t
            st = ctx.stats()
            dg, ent = loop_digest(r, owner)
            res.append((dg, ent, int(sum(x.info["total_kmers"] for x in r)), int(st["dropped_kmers"]), dt))
        del r
        for x in (packed, off, lens): H.pinned_free(x)
        ctx.close()
    ok = res[0][:3] == res[1][:3]
    fails += not ok
    if not ok:
        print("     entries %d / %d, total_kmers %d / %d, digests %s" % (res[0][1], res[1][1], res[0][2], res[1][2], "equal" if res[0][0] == res[1][0] else "differ"))
    print("%s seed %d K=%d EXT=%d L=%d U=%d RL=%d reads=%d cov=%.0f ntasks=%d ranks=%d rec_cap=%d homopolymers %s di %.0f%%: entries %d, dropped %d, %.0f ms against %.0f ms" %
          ("OK  " if ok else "FAIL", seed, K, EXT, L, U, RL, n, n * RL / G, ntasks, R, cap, {k: v for k, v in shares.items() if v}, di, res[0][1], res[0][3], res[0][4] * 1e3, res[1][4] * 1e3), flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
