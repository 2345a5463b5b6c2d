"""The multi-GPU data path on ONE GPU: virtual ranks (hsk_count_loopback) run probe -> dispatch ->
owner-grouped parse -> byte packing -> all-to-all-v plan -> multi-segment extraction -> sort -> count,
with device copies in place of RCCL.  Results must equal the reference's multi-rank outputs."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _split(H, seqs, R):
    counts = H.plan_partition_reads([len(s) for s in seqs], R) if R > 1 else np.array([len(seqs)])
    parts, first = [], 0
    for r in range(R):
        parts.append(seqs[first:first + int(counts[r])])
        first += int(counts[r])
    return parts


@pytest.mark.parametrize("R", [2, 3])
def test_loopback_equals_reference_multirank(R):
    import hysortk_amd as H
    seqs = util.read_fasta(util.GOLDEN + "/reads_small.fa")
    parts = _split(H, seqs, R)
    ntasks = H.plan_tot_tasks(2, R)                                   # the golden runs used 2 threads per rank
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=ntasks) as c:
        res, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in parts])
    assert len(owner) == ntasks and set(owner.tolist()) == set(range(R))
    lines = []
    for r, kl in enumerate(res):
        for t in range(ntasks):
            a, b = int(kl.task_off[t]), int(kl.task_off[t + 1])
            if owner[t] != r:
                assert a == b                                       # a rank only returns the tasks it owns
            else:
                seg = kl.kmers[a:b, 0]
                assert np.all(seg[1:] > seg[:-1])
        lines += ["%s\t%d" % (s, int(c_)) for s, c_ in zip(kl.strings(), kl.cnt)]
    gold = open(util.GOLDEN + "/count_k31_np%d.txt" % R).read().splitlines()
    assert sorted(lines) == gold
    histo = sum(np.pad(kl.histo, (0, 65536 - kl.histo.size)) for kl in res)
    assert H.histogram_text(histo) == util.load_json("dispatch_k31_np%d.json" % R)["histogram"]


@pytest.mark.parametrize("variant,R", [("k31ext", 2), ("k51", 4), ("k31", 8)])
def test_loopback_vs_oracle_with_owner_table(variant, R):
    """Per-rank lists equal the oracle restricted to the rank's tasks (incl. EXTENSION payload with global read ids)."""
    import hysortk_amd as H
    from hysortk_amd import synth
    from oracle import hsk_oracle as O
    cfg = util.VARIANTS[variant]
    seqs = synth.reads(60000, 150, 6000, 11)
    parts = _split(H, seqs, R)
    ntasks = 3 * R
    with H.Context(K=cfg["k"], M=cfg["m"], L=2, U=50, EXT=cfg["ext"], ntasks=ntasks) as c:
        res, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in parts])
    packed, off, lens = O.pack_reads(seqs)
    for r in range(R):
        ores = O.count(packed, off, lens, k=cfg["k"], m=cfg["m"], L=2, U=50, ext=cfg["ext"], ntasks=ntasks, task_owner=owner, my_rank=r)
        kl = res[r]
        assert np.array_equal(kl.kmers, ores.keys), r
        assert np.array_equal(kl.cnt, ores.cnt), r
        assert np.array_equal(kl.task_off, ores.task_off), r
        if cfg["ext"]:
            for i in range(0, len(kl), 53):
                pos, rid = kl.payload(i)
                a, b = int(ores.payoff[i]), int(ores.payoff[i + 1])
                assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(ores.rid[a:b].tolist(), ores.pos[a:b].tolist()))


@pytest.mark.parametrize("variant,R,ntasks", [("k31", 2, 40), ("k31ext", 3, 60), ("k51", 2, 34), ("k31", 8, 320)])     # (8, 320): the shape of the 8-GPU bench, 40 tasks = 5 groups per rank
def test_loopback_grouped_exchange(variant, R, ntasks):
    """Several task groups per rank: the exchange of group g+1 overlaps the sort of group g (GroupFeeder);
    per-rank lists must still equal the oracle restricted to the rank's tasks."""
    import hysortk_amd as H
    from hysortk_amd import synth
    from oracle import hsk_oracle as O
    cfg = util.VARIANTS[variant]
    seqs = synth.reads(150000, 150, 12000, 23)
    parts = _split(H, seqs, R)
    with H.Context(K=cfg["k"], M=cfg["m"], L=2, U=50, EXT=cfg["ext"], ntasks=ntasks) as c:
        res, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in parts])
    assert np.bincount(owner, minlength=R).max() > 8                  # more than one group on some rank
    packed, off, lens = O.pack_reads(seqs)
    for r in range(R):
        ores = O.count(packed, off, lens, k=cfg["k"], m=cfg["m"], L=2, U=50, ext=cfg["ext"], ntasks=ntasks, task_owner=owner, my_rank=r)
        kl = res[r]
        assert np.array_equal(kl.kmers, ores.keys), r
        assert np.array_equal(kl.cnt, ores.cnt), r
        assert np.array_equal(kl.task_off, ores.task_off), r
        if cfg["ext"]:
            for i in range(0, len(kl), 97):
                pos, rid = kl.payload(i)
                a, b = int(ores.payoff[i]), int(ores.payoff[i + 1])
                assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(ores.rid[a:b].tolist(), ores.pos[a:b].tolist()))


def test_overlap_switch_gives_identical_lists():
    """HSK_OVERLAP=0 (one exchange up front) and the grouped, overlapped exchange produce the same bytes."""
    import subprocess, sys, os
    code = ("import sys, numpy as np, hashlib; sys.path.insert(0, %r); import hysortk_amd as H\n"
            "from hysortk_amd import synth\n"
            "seqs = synth.reads(200000, 150, 9000, 3)\n"
            "parts = [seqs[:len(seqs)//2], seqs[len(seqs)//2:]]\n"
            "c = H.Context(K=31, M=17, L=1, U=80, ntasks=48)\n"
            "res, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in parts])\n"
            "h = hashlib.sha256(owner.tobytes())\n"
            "for kl in res: h.update(kl.kmers.tobytes() + kl.cnt.tobytes() + kl.task_off.tobytes() + kl.histo.tobytes())\n"
            "print(h.hexdigest(), sum(len(k) for k in res))\n") % util.ROOT
    outs = []
    # ... and so do the two supermer stores: bytes written by the placement (default when supermers travel) or positions + pack_kernel
    for env in ({"HSK_OVERLAP": "1"}, {"HSK_OVERLAP": "0"}, {"HSK_PLACE_BYTES": "0"}, {"HSK_PLACE_BYTES": "0", "HSK_OVERLAP": "0"}):
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **util.tune_env(env))).decode().split())
    assert all(o == outs[0] for o in outs), outs
    assert int(outs[0][1]) > 10000


@pytest.mark.gpu
def test_rccl_binding_selftest():
    """The RCCL entry points the exchange uses (dlopen'ed symbols, enum values, by-value unique id, grouped send/recv on the
    second stream) on a one-rank communicator: the only way to touch them on a single-GPU box."""
    import hysortk_amd as H
    with H.Context(K=31, M=17) as c:
        c.comm_selftest()


@pytest.mark.parametrize("K", [31, 51, 35, 77, -31, -51])
@pytest.mark.parametrize("R,ntasks", [(2, 24), (3, 24), (4, 32), (2, 56), (8, 128)])
def test_loopback_heavy_hitter_tasks(R, ntasks, K):
    """K = 31 / 51 / 35 / 77: one-, two- (with and without the prefix plan) and three-word keys (ScatteredKmerList is generic
    over TKmer, reference kmerops.cpp:363-401).  A tandem repeat makes a few tasks several times larger than the mean: they are classified as heavy hitters
    (reference HeavyHitterClassifier, kmerops.cpp:1157), every rank pre-aggregates its share into (k-mer, count) lists, the
    owner sums the lists (GatheredKmerList::process, kmerops.cpp:575).  Per-rank results must still equal the oracle."""
    import hysortk_amd as H
    from hysortk_amd import synth
    from oracle import hsk_oracle as O
    # K < 0: the same with the combining extraction planned on the owners' side (combine_min_bytes=0): the heavy tasks travel as lists, every
    # other task as supermers with their minimizer bits, and the owners build the items
    tuning = "combine_min_bytes=0,plan_sample=0" if K < 0 else None
    K = abs(K)
    rng = np.random.default_rng(5)
    seqs = synth.reads(80000, 150, 6000, 31)
    unit = "ACGGTCATTGCA"
    rep = (unit * 13)[:150]
    seqs = list(seqs) + [rep] * 2500 + [(unit[5:] + unit[:5]) * 12 + "ACGTAC"] * 500 + ["A" * 150] * 300    # tandem repeat + poly-A
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]                                  # every rank sees repeat copies
    parts = _split(H, seqs, R)
    if K != 31 and (R, ntasks) not in ((2, 24), (4, 32)):
        pytest.skip("the wider keys take two of the rank / task layouts")
    with H.Context(K=K, M=17, L=2, U=65535, ntasks=ntasks, tuning=tuning) as c:
        res, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in parts])
        st = c.stats()
    assert st["heavy_tasks"] > 0, st
    assert (st["combine_pairs"] > 0) == (tuning is not None), st
    packed, off, lens = O.pack_reads(seqs)
    total = 0
    for r in range(R):
        ores = O.count(packed, off, lens, k=K, m=17, L=2, U=65535, ntasks=ntasks, task_owner=owner, my_rank=r)
        kl = res[r]
        assert np.array_equal(kl.task_off, ores.task_off), r
        assert np.array_equal(kl.kmers, ores.keys), r
        assert np.array_equal(kl.cnt, ores.cnt), r
        assert H.histogram_text(kl.histo) == O.histogram_text(ores.cnt)
        total += len(kl)
    assert total > 1000 and max(int(kl.cnt.max()) for kl in res if len(kl)) > 1000    # the repeat's k-mers made it through the merge
    # (round 4's fuzz: the heavy tasks' instances arrive as lists, not as supermers -- their owners left them out of total_kmers)
    assert sum(int(kl.info["total_kmers"]) for kl in res) == sum(len(s) - K + 1 for s in seqs)


# ---- the 8-GPU configurations' per-GPU share through the multi-rank data path at full size, on one GPU ---------------------------
def _loopback_full(H, K, EXT, R, G, NR, ntasks, seed=20251003):
    """R virtual ranks, each holding NR reads sampled from ONE genome of G bases (bench.py's N > 1 workload): per-rank results in
    host memory + the owner table + per-task oracle digests over ALL reads (global read ids)."""
    from oracle import hsk_oracle as O
    RL = 150
    want_n = np.zeros(ntasks, np.uint64); want_mix = np.zeros(ntasks, np.uint64)
    with H.Context(K=K, M=17, L=1, U=65535, EXT=EXT, ntasks=ntasks, profile=True) as c:
        reads = []
        for r in range(R):
            dp, nb, do, dl = c.synth_reads(G, RL, NR, seed, first_read=r * NR)
            reads.append((dp, nb, do, dl, NR))
        res, owner = c.count_loopback_device(reads)
        st = c.stats()
        off = np.arange(NR, dtype=np.uint64) * np.uint64((RL + 3) // 4)
        lens = np.full(NR, RL, dtype=np.uint32)
        for r in range(R):
            packed = c.d2h(reads[r][0], reads[r][1])
            n, mix = O.task_digests(packed, off, lens, k=K, m=17, ext=EXT, ntasks=ntasks, rid_base=r * NR)
            with np.errstate(over="ignore"):
                want_n += n; want_mix += mix
            del packed
        for x in reads:
            c.synth_free(x[0], x[2], x[3])
    return res, owner, want_n, want_mix, st


def _check_ranks_against_digests(res, owner, want_n, want_mix, R, nw, ext):
    from oracle import hsk_oracle as O
    ntasks = len(owner)
    seen = np.zeros(ntasks, bool)
    for r, kl in enumerate(res):
        pay = (kl.payload_off, kl.pos, kl.rid) if ext else None
        got_n, got_mix = O.entries_digests(kl.kmers, kl.cnt, kl.task_off, payload=pay)
        mine = owner == r
        assert not np.any(got_n[~mine]), "a rank returned entries of a task it does not own"
        assert np.array_equal(got_n[mine], want_n[mine]), (r, got_n[mine], want_n[mine])
        assert np.array_equal(got_mix[mine], want_mix[mine]), r
        seen |= mine
        # every task strictly ascending as a little-endian multi-word integer
        k = kl.kmers
        up = k[1:, nw - 1] > k[:-1, nw - 1]
        for w in range(nw - 2, -1, -1):
            eq = np.ones(len(up), bool)
            for w2 in range(nw - 1, w, -1):
                eq &= k[1:, w2] == k[:-1, w2]
            up |= eq & (k[1:, w] > k[:-1, w])
        starts = kl.task_off[1:-1].astype(np.int64)
        up[starts[(starts > 0) & (starts < len(k))] - 1] = True
        assert bool(up.all()), r
    assert seen.all() and set(owner.tolist()) == set(range(R))


@pytest.mark.parametrize("K,EXT", [(31, 0), (51, 0), (31, 1)])
def test_full_size_multirank_path_eight_virtual_ranks(K, EXT):
    """BASELINE configs[2] / [3] / [4] need 8 GPUs; what ONE GPU can prove of them: 8 virtual ranks x 1.0 Gbp of reads sampled from one
    genome (8 Gbp: the oracle's streaming on the box's 16 host threads is what the test's time goes into), the 8-GPU bench's 320 tasks (40 per rank = 5 task groups each), dispatcher, byte-store
    placement, grouped exchange overlapped with the sort (device copies in place of RCCL send / recv), multi-segment extraction from
    eight source ranks per task: every rank's list equals the CPU oracle's per-task digests of ALL reads (count and multiset digest,
    with EXTENSION every (k-mer, pos, global read id)), every task strictly ascending, key arrays beyond 2^32 bytes."""
    import hysortk_amd as H
    R, G, RL = 8, (250_000_000 if not EXT else 39_062_500), 150          # (8 x 1.0 Gbp; EXTENSION: 1.0e9 payloads of 8 bytes come back to the host; 1.25 Gbp in all)
    NR = G * 32 // RL // R
    res, owner, want_n, want_mix, st = _loopback_full(H, K, EXT, R, G, NR, 320)
    assert int(want_n.sum()) == R * NR * (RL - K + 1) == sum(kl.info["total_kmers"] for kl in res)
    _check_ranks_against_digests(res, owner, want_n, want_mix, R, (K + 31) // 32, EXT)
    assert np.bincount(owner, minlength=R).min() >= 30                  # (balanced dispatch: ~40 tasks per rank)
    # the plan is chosen from ONE rank's sketch scaled to the whole input (each rank holds 4-fold coverage of its own, the job 32-fold):
    # the owner-side combining extraction must be what ran, for one- and two-word keys; payloads never take it
    assert (st["combine_launches"] > 0) == (not EXT), st


def test_full_size_multirank_path_byte_store_beyond_4gb():
    """Two virtual ranks x 5 Gbp: each rank's byte store (the exchange's wire format: 6.2 GB) and receive buffers pass 2^32 bytes, a task's
    segments from both source ranks lie beyond 32-bit offsets; 96 tasks = 6 task groups per rank.  (Rounds 3 - 4 ran it with 2 x 12 Gbp: 74 s of a
    suite that has to stay inside the driver's window; the offsets cross 2^32 either way.)"""
    import hysortk_amd as H
    R, G, RL = 2, 312_500_000, 150
    NR = 5_000_000_000 // RL
    res, owner, want_n, want_mix, st = _loopback_full(H, 31, 0, R, G, NR, 96)
    assert sum(kl.info["total_kmers"] for kl in res) == R * NR * (RL - 31 + 1)
    _check_ranks_against_digests(res, owner, want_n, want_mix, R, 1, 0)


def test_multirank_path_with_more_than_eight_task_groups_per_rank():
    """704 tasks on 8 virtual ranks = 88 per rank = 11 task groups each (the group feeder keeps two in flight): lists equal the oracle's."""
    import hysortk_amd as H
    R, G, RL = 8, 40_000_000, 150
    NR = G * 32 // RL // R
    res, owner, want_n, want_mix, st = _loopback_full(H, 31, 0, R, G, NR, 704)
    _check_ranks_against_digests(res, owner, want_n, want_mix, R, 1, 0)
    assert np.bincount(owner, minlength=R).min() > 64
