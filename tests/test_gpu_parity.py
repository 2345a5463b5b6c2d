"""Parity of the HIP path (through the C ABI of libhsk.so) against the oracle and the golden
fixtures produced by the real reference.  Bit-exact: everything here is integer/byte work."""
import io

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import hsk_oracle
    return hsk_oracle


@pytest.fixture(scope="module")
def H():
    import hysortk_amd
    return hysortk_amd


def _ctx(H, variant, **kw):
    cfg = util.VARIANTS[variant]
    args = dict(K=cfg["k"], M=cfg["m"], L=cfg["L"], U=cfg["U"], EXT=cfg["ext"])
    args.update(kw)
    return H.Context(**args)


def _small_reads(H):
    seqs = util.read_fasta(util.GOLDEN + "/reads_small.fa")
    return seqs, H.DnaBuffer.from_sequences(seqs)


# ---------------------------------------------------------------------------------------------------
# a1: packing (host code of the shim, checked here against the reference's bytes as well)
# ---------------------------------------------------------------------------------------------------
def test_pack_matches_reference_bytes(H):
    g = util.load_json("stages_k31.json")
    for rd in g["reads"]:
        assert H.pack_sequence(rd["seq"]).tobytes().hex() == rd["packed"]


# ---------------------------------------------------------------------------------------------------
# a4: destinations
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,tot", [("k31", 5), ("k31", 47), ("k51", 5), ("k21", 47), ("k51m35", 5), ("k51m35", 47), ("k77m65", 47)])
def test_stage_destinations_golden(H, variant, tot):
    g = util.load_json("stages_%s.json" % variant)
    seqs = [rd["seq"] for rd in g["reads"]]
    dna = H.DnaBuffer.from_sequences(seqs)
    with _ctx(H, variant, ntasks=tot) as c:
        dest, doff = c.stage_destinations(dna)
    for r, rd in enumerate(g["reads"]):
        want = rd["tasks"][str(tot)]["dest"]
        got = dest[int(doff[r]):int(doff[r + 1])].tolist()
        assert got == want, "read %d" % r


def test_stage_destinations_vs_oracle_reads(H, O):
    seqs, dna = _small_reads(H)
    packed, off, lens = dna.arrays()
    with _ctx(H, "k31", ntasks=13) as c:
        dest, doff = c.stage_destinations(dna)
    for r in range(len(seqs)):
        want = O.dests(packed[int(off[r]):], int(lens[r]), 31, 17, 13)
        assert np.array_equal(dest[int(doff[r]):int(doff[r + 1])], want), r


# ---------------------------------------------------------------------------------------------------
# a5 + a11: per-task canonical k-mers (supermer split + extraction); order unspecified
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["k31", "k51", "k21", "k31ext"])
def test_stage_task_kmers_vs_oracle(H, O, variant):
    cfg = util.VARIANTS[variant]
    seqs, dna = _small_reads(H)
    packed, off, lens = dna.arrays()
    ntasks = 7
    ores = O.count(packed, off, lens, k=cfg["k"], m=cfg["m"], L=1, U=65535, ext=cfg["ext"], ntasks=ntasks, rid_base=100)
    with _ctx(H, variant, ntasks=ntasks) as c:
        for t in range(ntasks):
            keys, pos, rid = c.stage_task_kmers(dna, t, rid_base=100)
            a, b = int(ores.task_off[t]), int(ores.task_off[t + 1])
            okeys = np.repeat(ores.keys[a:b], ores.cnt[a:b].astype(np.int64), axis=0)
            assert keys.shape == okeys.shape
            if cfg["ext"]:
                pa, pb = int(ores.payoff[a]), int(ores.payoff[b])
                got = sorted(zip(map(tuple, keys.tolist()), rid.tolist(), pos.tolist()))
                want = sorted(zip(map(tuple, okeys.tolist()), ores.rid[pa:pb].tolist(), ores.pos[pa:pb].tolist()))
                assert got == want
            else:
                order = np.lexsort(keys.T[::-1])
                oorder = np.lexsort(okeys.T[::-1])
                assert np.array_equal(keys[order], okeys[oorder])


# ---------------------------------------------------------------------------------------------------
# a12: sort_task
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nw", [1, 2, 3])
@pytest.mark.parametrize("n", [1, 2, 63, 4096, 4097, 100003, 1 << 20])
def test_stage_sort(H, nw, n):
    rng = np.random.default_rng(n * 7 + nw)
    keys = rng.integers(0, 1 << 63, size=(n, nw), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, nw), dtype=np.uint64)
    if n > 100:
        keys[: n // 3] = keys[n // 3: 2 * (n // 3)]          # plenty of duplicates
        keys[::7, nw - 1] &= np.uint64(0xFF)                  # skewed high digits
    with H.Context(K=31 if nw == 1 else (51 if nw == 2 else 80)) as c:
        out = c.stage_sort(keys)
        vals = np.arange(n, dtype=np.uint64)
        out2, v2 = c.stage_sort(keys, vals)
    # little-endian multiword order: word nw-1 most significant
    order = np.lexsort([keys[:, w] for w in range(nw)])
    assert np.array_equal(out, keys[order])
    assert np.array_equal(out2, keys[order])
    assert np.array_equal(v2, vals[order])                    # LSD radix is stable, lexsort too


def test_stage_sort_all_equal_and_sorted_inputs(H):
    with H.Context() as c:
        k = np.full(10000, 0x1234567800000000, dtype=np.uint64)
        assert np.array_equal(c.stage_sort(k).reshape(-1), k)
        k = np.arange(50000, dtype=np.uint64) * np.uint64(0x100000001)
        assert np.array_equal(c.stage_sort(k[::-1].copy()).reshape(-1), k)


# ---------------------------------------------------------------------------------------------------
# a13: count_sorted_kmers
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,U", [(1, 65535), (2, 5), (15, 40)])
@pytest.mark.parametrize("nw", [1, 2])
def test_stage_count_sorted(H, L, U, nw):
    rng = np.random.default_rng(L * 100 + U + nw)
    distinct = np.unique(rng.integers(0, 1 << 62, size=(3000, nw), dtype=np.uint64), axis=0)
    reps = rng.integers(1, 60, size=distinct.shape[0])
    reps[5] = 70000                                           # a run longer than U and than a tile
    reps[17] = 5000
    keys = np.repeat(distinct, reps, axis=0)
    order = np.lexsort([keys[:, w] for w in range(nw)])
    keys = keys[order]
    with H.Context(K=31 if nw == 1 else 51, L=L, U=U) as c:
        k, cnt = c.stage_count_sorted(keys)
    u, ucnt = np.unique(keys, axis=0, return_counts=True)
    uo = np.lexsort([u[:, w] for w in range(nw)])
    u, ucnt = u[uo], ucnt[uo]
    keep = (ucnt >= L) & (ucnt <= U)
    assert np.array_equal(k, u[keep])
    assert np.array_equal(cnt, ucnt[keep].astype(np.uint64))


# ---------------------------------------------------------------------------------------------------
# the whole path against the reference's own output
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["k31", "k31f", "k21", "k51", "k51m35", "k77m65"])
def test_count_golden_raw_order(H, variant):
    """ntasks = 5 (what the reference used: 1 rank x 8 threads): the raw KmerListS is reproduced
    element for element, and so is the printed histogram."""
    cfg = util.VARIANTS[variant]
    _, dna = _small_reads(H)
    with _ctx(H, variant, ntasks=5) as c:
        res = c.count(dna)
    gold = util.load_count("count_%s.txt" % variant)
    assert res.strings() == [g[0] for g in gold]
    assert res.cnt.tolist() == [g[1] for g in gold]
    assert H.histogram_text(res.histo) == open(util.GOLDEN + "/hist_%s.txt" % variant).read()
    buf = io.StringIO()
    H.print_kmer_histogram(res, file=buf)
    assert buf.getvalue() == open(util.GOLDEN + "/hist_%s.txt" % variant).read()


def test_count_golden_k51_paradis_multiset(H):
    _, dna = _small_reads(H)
    with _ctx(H, "k51p", ntasks=5) as c:
        res = c.count(dna)
    gold = util.load_count("count_k51p.txt")
    assert sorted(zip(res.strings(), res.cnt.tolist())) == sorted((g[0], g[1]) for g in gold)
    # ... and element for element once the list is put into PARADIS order (unfiltered lists only, see hysortk_amd.paradis_order)
    perm = H.paradis_order(res.kmers, res.cnt, res.task_off)
    strs = res.strings()
    assert [strs[i] for i in perm] == [g[0] for g in gold] and res.cnt[perm].tolist() == [g[1] for g in gold]


def test_count_golden_extension(H):
    _, dna = _small_reads(H)
    with _ctx(H, "k31ext", ntasks=5) as c:
        res = c.count(dna)
    gold = util.load_count("count_k31ext.txt")
    assert res.strings() == [g[0] for g in gold]
    assert res.cnt.tolist() == [g[1] for g in gold]
    for i, g in enumerate(gold):
        pos, rid = res.payload(i)
        assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(g[3], g[2])), g[0]


@pytest.mark.parametrize("ntasks", [1, 2, 9, 64, 0])
def test_count_any_task_count_same_multiset(H, ntasks):
    """The content must not depend on tot_tasks (SURVEY 8a ordering contract)."""
    _, dna = _small_reads(H)
    with _ctx(H, "k31", ntasks=ntasks) as c:
        res = c.count(dna)
    gold = util.load_count("count_k31.txt")
    assert sorted(zip(res.strings(), res.cnt.tolist())) == sorted((g[0], g[1]) for g in gold)
    # per-task ascending runs
    for t in range(res.info["ntasks"]):
        seg = res.kmers[int(res.task_off[t]):int(res.task_off[t + 1]), 0]
        assert np.all(seg[1:] > seg[:-1])


def test_write_output_file(H, tmp_path):
    _, dna = _small_reads(H)
    with _ctx(H, "k31", ntasks=5) as c:
        res = c.count(dna)
    H.write_output_file(res, str(tmp_path))
    lines = open(tmp_path / "0.out").read().splitlines()
    gold = util.load_count("count_k31.txt")
    assert lines == ["%s\t%d" % (g[0], g[1]) for g in gold]


# ---------------------------------------------------------------------------------------------------
# edge cases the reference handles (or trips over)
# ---------------------------------------------------------------------------------------------------
def test_edge_cases(H, O):
    cases = {
        "empty": [],
        "all_short": ["ACGT", "A" * 30, ""],
        "exactly_k": ["ACGTTGCAAGGCTTAACCGGTTACGATCGAT"],
        "with_n": ["ACGTNNNNACGTTGCAAGGCTTAACCGGTTACGATCGATNACGTTGCA"],
        "poly_a": ["A" * 700],
        "mixed_empty": ["", "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGGCTAAGC", "", "", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", ""],
        "one_byte_reads": ["A", "C", "G", "T"] * 50 + ["ACGTTGCAAGGCTTAACCGGTTACGATCGATCG"],
    }
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=3) as c:
        for name, seqs in cases.items():
            dna = H.DnaBuffer.from_sequences(seqs)
            res = c.count(dna)
            packed, off, lens = O.pack_reads(seqs)
            ores = O.count(packed, off, lens, k=31, m=17, L=1, U=65535, ntasks=3)
            assert res.kmers.shape == ores.keys.shape, name
            assert np.array_equal(res.kmers, ores.keys), name
            assert np.array_equal(res.cnt, ores.cnt), name
            assert np.array_equal(res.task_off, ores.task_off), name
            assert res.info["total_kmers"] == ores.stats["total_kmers"], name


def _random_record(seed, n):
    codes = np.random.Generator(np.random.PCG64(seed)).integers(0, 4, size=n, dtype=np.uint8)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes().decode()


@pytest.mark.parametrize("ntasks", [5, 0])
def test_s_ecoli_single_record(H, O, ntasks):
    """BASELINE.json configs[0] (S-ecoli stand-in): ONE record of 4 641 652 uniform random bases
    (PCG64 seed 42), K=31 M=17 L=1 U=65535 -> N = 4 641 622 k-mers, every one of them once: histogram text
    "#count\tnumkmers\n1\t4641622\n\n" as the reference prints it (src/hysortk.cpp:122-131).  One read spans ~2270 parse
    tiles and ~9000 supermer cuts; ntasks = 5 is the reference's own task count for 1 rank x 8 threads."""
    seq = _random_record(42, 4_641_652)
    dna = H.DnaBuffer.from_sequences([seq])
    packed, off, lens = dna.arrays()
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=ntasks) as c:
        res = c.count(dna)
    assert res.info["total_kmers"] == 4_641_622
    assert H.histogram_text(res.histo) == "#count\tnumkmers\n1\t4641622\n\n"
    assert len(res) == 4_641_622 and int(res.cnt.sum()) == 4_641_622
    ores = O.count(packed, off, lens, k=31, m=17, L=1, U=65535, ntasks=res.info["ntasks"], fast=True)
    assert np.array_equal(res.task_off, ores.task_off)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)


@pytest.mark.parametrize("ntasks", [1, 5, 40])
def test_long_records_and_tile_edges(H, O, ntasks):
    """Mbp-long records: a record that ends exactly on a 2048-position parse tile edge (the next one starts on it), a 1 Mbp
    record, one that is a repeat of the first (counts of 2), a record shorter than K between them, and a last record that
    ends exactly at the end of a tile; single-task path, padded batch and full batches."""
    a = _random_record(7, 2048 * 489)                       # 1 001 472 bases: ends on a tile edge (records are byte-aligned)
    b = _random_record(8, 1_000_000)
    reads = [a, b, "ACGTACGTACGT", a[1000:300000]]
    pre = sum((len(r) + 3) // 4 for r in reads)             # bytes in front of the last record
    reads.append(_random_record(9, 4 * ((-pre) % 512 + 1024)))
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    assert (int(off[1]) * 4) % 2048 == 0 and (packed.size * 4) % 2048 == 0
    ores = O.count(packed, off, lens, k=31, m=17, L=1, U=65535, ntasks=ntasks, fast=True)
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=ntasks) as c:
        res = c.count(dna)
    assert res.info["total_kmers"] == sum(max(0, len(r) - 30) for r in reads)
    assert np.array_equal(res.task_off, ores.task_off)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)
    assert int(res.histo[2]) >= 299000 - 30


def test_invalid_config_is_rejected(H):
    for kw in (dict(K=32), dict(K=2), dict(K=96), dict(M=31, K=31), dict(M=32, K=51), dict(M=64, K=77), dict(L=0), dict(L=5, U=4), dict(U=70000), dict(EXT=2)):
        with pytest.raises(H.HskError):
            H.Context(**kw)


# ---------------------------------------------------------------------------------------------------
# synthetic reads: device generator == numpy twin; path == oracle on them
# ---------------------------------------------------------------------------------------------------
def test_synth_device_equals_numpy(H):
    from hysortk_amd import synth
    with H.Context() as c:
        dp, nb, do, dl = c.synth_reads(50000, 150, 3000, 42)
        got = c.d2h(dp, nb)
        c.synth_free(dp, do, dl)
    packed, off, lens = synth.packed_reads(50000, 150, 3000, 42)
    assert np.array_equal(got, packed)
    seqs = synth.reads(50000, 150, 40, 42)
    nbr = (150 + 3) // 4
    for r, s in enumerate(seqs):
        assert H.pack_sequence(s).tobytes() == packed[r * nbr:(r + 1) * nbr].tobytes()


@pytest.mark.parametrize("variant,EXT", [("k31", 0), ("k31", 1), ("k51", 0)])
def test_count_synth_vs_oracle(H, O, variant, EXT):
    from hysortk_amd import synth
    cfg = util.VARIANTS[variant]
    packed, off, lens = synth.packed_reads(200000, 150, 20000, 7)      # 15x of a 200 kbp genome, 2.4 M k-mers
    ores = O.count(packed, off, lens, k=cfg["k"], m=cfg["m"], L=2, U=40, ext=EXT, ntasks=6, fast=True)
    with H.Context(K=cfg["k"], M=cfg["m"], L=2, U=40, EXT=EXT, ntasks=6) as c:
        dp, nb, do, dl = c.synth_reads(200000, 150, 20000, 7)
        res = c.count_device(dp, nb, do, dl, 20000)
        c.synth_free(dp, do, dl)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)
    assert np.array_equal(res.task_off, ores.task_off)
    assert res.info["total_kmers"] == ores.stats["total_kmers"] == 20000 * (150 - cfg["k"] + 1)
    if EXT:
        # payload order inside a k-mer is unspecified (as in the reference): compare as sets per k-mer
        for i in list(range(0, len(res), 97)) + [len(res) - 1]:
            pos, rid = res.payload(i)
            a, b = int(ores.payoff[i]), int(ores.payoff[i + 1])
            assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(ores.rid[a:b].tolist(), ores.pos[a:b].tolist()))
        # every kept (k-mer, rid, pos) triple, all entries at once
        sel = np.concatenate([np.arange(int(o), int(o) + int(c)) for o, c in zip(res.payload_off[:-1], res.cnt)])
        assert sorted(zip(res.rid[sel].tolist(), res.pos[sel].tolist())) == sorted(zip(ores.rid.tolist(), ores.pos.tolist()))


def test_count_reads_with_errors_vs_oracle(H, O):
    """1 % substitution errors (hsk_synth_reads_err): most erroneous k-mers are singletons, prefix bins hold several times
    more distinct keys than with error-free reads, so the aggregation's small table overflows in places and the retry with
    the large table (or the long way) takes over; the list must still be the oracle's."""
    G, RL, NR = 400000, 150, 80000                              # 30x
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=8) as c:
        dp, nb, do, dl = c.synth_reads(G, RL, NR, 13, error_rate=0.01)
        packed = c.d2h(dp, nb)
        res = c.count_device(dp, nb, do, dl, NR)
        c.synth_free(dp, do, dl)
        st = c.stats()
    from hysortk_amd import synth
    clean, off, lens = synth.packed_reads(G, RL, NR, 13)
    diff = np.unpackbits(packed ^ clean).reshape(-1, 2).any(axis=1).mean()
    assert 0.007 < diff < 0.013                                  # about one base in a hundred differs from the error-free twin
    ores = O.count(packed, off, lens, k=31, m=17, L=1, U=65535, ntasks=8, fast=True)
    assert st["fused_tasks"] + st["redone_tasks"] == 8
    assert np.array_equal(res.task_off, ores.task_off) and np.array_equal(res.kmers, ores.keys) and np.array_equal(res.cnt, ores.cnt)
    assert int(res.histo[1]) > 5 * int(res.histo[2])             # the error tail: singletons dominate


# ---------------------------------------------------------------------------------------------------
# size-independent properties at a size the oracle does not run in seconds
# ---------------------------------------------------------------------------------------------------
def test_large_properties(H):
    G, RL, NR = 20_000_000, 150, 4_000_000            # 600 Mbp, 480 M k-mers, 30x
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=0) as c:
        dp, nb, do, dl = c.synth_reads(G, RL, NR, 99)
        res = c.count_device(dp, nb, do, dl, NR)
        c.synth_free(dp, do, dl)
    total = NR * (RL - 31 + 1)
    assert res.info["total_kmers"] == total
    assert int(res.cnt.sum()) == total                                  # checksum of counts (L=1: nothing filtered)
    assert int((res.histo * np.arange(res.histo.size, dtype=np.uint64)).sum()) == total
    assert int(res.histo.sum()) == len(res)
    for t in range(res.info["ntasks"]):                                 # sortedness / uniqueness inside every task
        seg = res.kmers[int(res.task_off[t]):int(res.task_off[t + 1]), 0]
        assert np.all(seg[1:] > seg[:-1])
    assert np.unique(res.kmers[:, 0]).size == len(res)                  # no k-mer in two tasks
    assert np.all((res.kmers[:, 0] & np.uint64(3)) == 0)                # K=31: two low bits unused
    # idempotence: counting again gives the identical list
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=0) as c:
        dp, nb, do, dl = c.synth_reads(G, RL, NR, 99)
        res2 = c.count_device(dp, nb, do, dl, NR)
        c.synth_free(dp, do, dl)
    assert np.array_equal(res.kmers, res2.kmers) and np.array_equal(res.cnt, res2.cnt)


def test_full_size_properties_both_expand_paths():
    """BASELINE.json configs[1] at full size (10 Gbp, 8.0e9 31-mers, 40 tasks of 2e8 k-mers): size-independent properties of the
    unfiltered list -- checksum of counts = number of k-mers, histogram consistent, every task strictly ascending, low bits
    clear -- and a checksum of checksums that must be identical with the expand fused into the first scatter pass (default)
    and with expand + two passes (HSK_FUSED_SCATTER=0).  Subprocesses: the switch is read once, and each run holds 5 GB."""
    import subprocess, sys, os
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import hysortk_amd as H\n"
            "G, RL = 312500000, 150; NR = G * 32 // RL\n"
            "c = H.Context(K=31, M=17, L=1, U=65535, ntasks=0)\n"
            "dp, nb, do, dl = c.synth_reads(G, RL, NR, 20251003)\n"
            "r = c.count_device(dp, nb, do, dl, NR)\n"
            "total = NR * (RL - 31 + 1)\n"
            "assert r.info['total_kmers'] == total\n"
            "k = r.kmers[:, 0]; cnt = r.cnt\n"
            "assert int(cnt.sum(dtype=np.uint64)) == total\n"
            "assert int((r.histo * np.arange(r.histo.size, dtype=np.uint64)).sum()) == total and int(r.histo.sum()) == len(k)\n"
            "assert not np.any(k & np.uint64(3))\n"
            "d = k[1:] > k[:-1]; starts = r.task_off[1:-1].astype(np.int64)\n"
            "d[starts[(starts > 0) & (starts < len(k))] - 1] = True\n"
            "assert bool(d.all())\n"
            "h = int(np.bitwise_xor.reduce(k * np.uint64(0x9E3779B97F4A7C15) + cnt.astype(np.uint64)))\n"
            "print(len(k), h, int(k.sum(dtype=np.uint64)), r.info['ntasks'], c.stats()['fused_tasks'])\n") % util.ROOT
    outs = [subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **util.tune_env(env))).decode().split() for env in ({}, {"HSK_FUSED_SCATTER": "0"})]
    assert outs[0] == outs[1], outs
    assert int(outs[0][0]) > 300_000_000 and int(outs[0][3]) == 40 and int(outs[0][4]) == 40


def test_reported_supermer_totals_follow_the_documented_rule(H, O):
    """hsk_result.total_supermers / total_supermer_bytes (the 1.1 B per k-mer of the roofline arithmetic) against an independent
    restatement of THIS library's supermer rule (a supermer starts where a read's first k-mer starts, at every
    position of the rank's base stream that is a multiple of 128, and wherever the window minimum -- the hash VALUE -- changes),
    computed from the oracle's m-mer hashes.  The k-mers those supermers hold are compared elsewhere (per-task multisets)."""
    from hysortk_amd import synth
    for K, M, nreads, rl in ((31, 17, 3000, 150), (51, 17, 1500, 250), (21, 9, 2000, 101), (31, 17, 40, 5000)):
        seqs = list(synth.reads(60000, rl, nreads, 17)) + ["ACGT" * 40, "A" * 300, "C" * (K - 1), "G" * K]
        pk, off, ln = O.pack_reads(seqs)
        nsup = nbytes = nsup16 = nbytes16 = 0
        W = K - M + 1
        for r, s_ in enumerate(seqs):
            n = len(s_)
            if n < K:
                continue
            h = O.mmer_hashes(pk[int(off[r]):], n, M)
            mins = np.lib.stride_tricks.sliding_window_view(h, W).min(axis=1)         # one per k-mer
            g = 4 * int(off[r]) + np.arange(mins.size)
            start = np.ones(mins.size, dtype=bool)
            start[1:] = (mins[1:] != mins[:-1]) | (g[1:] % 128 == 0)
            idx = np.flatnonzero(start)
            nk = np.diff(np.append(idx, mins.size))
            nsup += idx.size
            nbytes += int(((nk + K - 1 + 3) // 4).sum())
            # the combining extraction (hsk_combine.h) cuts a run again every 16 k-mers from its start: one 16-byte item per supermer
            pieces = np.concatenate([np.full(n_ // 16, 16, dtype=np.int64) if n_ % 16 == 0 else np.append(np.full(n_ // 16, 16, dtype=np.int64), n_ % 16) for n_ in nk])
            nsup16 += pieces.size
            nbytes16 += int(((pieces + K - 1 + 3) // 4).sum())
        with H.Context(K=K, M=M, L=1, U=65535, ntasks=7) as c:
            res = c.count((pk, off, ln))
            combined = c.stats()["combine_pairs"] > 0            # (only with HSK_COMBINE_MIN_BYTES=0: inputs of this size take the instance path)
        want, wantb = (nsup16, nbytes16) if combined else (nsup, nbytes)
        assert res.info["total_supermers"] == want, (K, M, res.info["total_supermers"], nsup, nsup16)
        assert res.info["total_supermer_bytes"] == wantb + want, (K, M)
        assert int(nk.max()) <= 128


def _oracle_digests_of_device_reads(c, dp, nb, do, dl, NR, RL, k, ext, ntasks, rid_base=0):
    """per-task (n, mix) of the reads resident in HBM, computed by the CPU oracle's streaming digests (oracle/hsk_oracle.c
    hsko_task_digests: every k-mer instance of every read, ~20 M positions/s per host thread)"""
    from oracle import hsk_oracle as O
    packed = c.d2h(dp, nb)
    off = np.arange(NR, dtype=np.uint64) * np.uint64((RL + 3) // 4)
    lens = np.full(NR, RL, dtype=np.uint32)
    return O.task_digests(packed, off, lens, k=k, m=17, ext=ext, ntasks=ntasks, rid_base=rid_base)


def test_full_size_equals_oracle_digests_k31(H):
    """BASELINE.json configs[1] at FULL size against the oracle, not only against itself: for each of the 40 tasks the number of
    k-mers and the multiset digest of the list (sum of cnt * mix(key) mod 2^64) equal what the CPU oracle streams out of the
    same 10 Gbp of reads (8.0e9 k-mer instances, record offsets far beyond 2^32).  With every task strictly ascending (checked
    here too) equal digests mean the list IS the task's k-mer multiset."""
    from oracle import hsk_oracle as O
    G, RL = 312_500_000, 150
    NR = G * 32 // RL
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=0) as c:
        dp, nb, do, dl = c.synth_reads(G, RL, NR, 20251003)
        r = c.count_device(dp, nb, do, dl, NR)
        nt = r.info["ntasks"]
        want_n, want_mix = _oracle_digests_of_device_reads(c, dp, nb, do, dl, NR, RL, 31, 0, nt)
        c.synth_free(dp, do, dl)
    assert nt == 40 and int(want_n.sum()) == NR * (RL - 31 + 1) == r.info["total_kmers"]
    got_n, got_mix = O.entries_digests(r.kmers, r.cnt, r.task_off)
    assert np.array_equal(got_n, want_n), (got_n, want_n)
    assert np.array_equal(got_mix, want_mix)
    k = r.kmers[:, 0]
    d = k[1:] > k[:-1]; starts = r.task_off[1:-1].astype(np.int64)
    d[starts[(starts > 0) & (starts < len(k))] - 1] = True
    assert bool(d.all())


def test_full_size_properties_k51(H):
    """BASELINE.md section 3, second record shape at full size: the 10 Gbp of reads at K=51 (two-word keys, 6.67e9 51-mers,
    unfiltered).  Size-independent properties: checksum of counts = number of k-mers, histogram consistent, every task strictly
    ascending as a little-endian two-word integer (RADULS order), unused low bits clear, no k-mer in two tasks (sampled), and the
    list is reproduced exactly by a second run."""
    G, RL = 312_500_000, 150
    NR = G * 32 // RL
    total = NR * (RL - 51 + 1)
    sums = []
    for _ in range(2):
        with H.Context(K=51, M=17, L=1, U=65535, ntasks=0) as c:
            dp, nb, do, dl = c.synth_reads(G, RL, NR, 20251003)
            r = c.count_device(dp, nb, do, dl, NR)
            c.synth_free(dp, do, dl)
            st = c.stats()
        assert r.info["total_kmers"] == total
        w0, w1, cnt = r.kmers[:, 0], r.kmers[:, 1], r.cnt
        assert int(cnt.sum(dtype=np.uint64)) == total
        assert int((r.histo * np.arange(r.histo.size, dtype=np.uint64)).sum()) == total and int(r.histo.sum()) == len(cnt)
        assert not np.any(w1 & np.uint64((1 << 26) - 1))                 # K=51: 19 bases in word 1, 26 low bits unused
        up = (w1[1:] > w1[:-1]) | ((w1[1:] == w1[:-1]) & (w0[1:] > w0[:-1]))
        starts = r.task_off[1:-1].astype(np.int64)
        up[starts[(starts > 0) & (starts < len(cnt))] - 1] = True
        assert bool(up.all())
        assert st["fused_tasks"] + st["redone_tasks"] == r.info["ntasks"]
        h = np.bitwise_xor.reduce((w0 * np.uint64(0x9E3779B97F4A7C15)) ^ (w1 * np.uint64(0xC2B2AE3D27D4EB4F)) ^ cnt)
        sums.append((len(cnt), int(h), int(w0.sum(dtype=np.uint64)), r.info["ntasks"]))
        if len(sums) == 1:
            # ... and against the CPU oracle: per task, number of 51-mers and multiset digest of the list (6.67e9 instances streamed on the host)
            from oracle import hsk_oracle as O
            with H.Context(K=51, M=17) as c2:
                dp, nb, do, dl = c2.synth_reads(G, RL, NR, 20251003)
                want_n, want_mix = _oracle_digests_of_device_reads(c2, dp, nb, do, dl, NR, RL, 51, 0, r.info["ntasks"])
                c2.synth_free(dp, do, dl)
            got_n, got_mix = O.entries_digests(r.kmers, r.cnt, r.task_off)
            assert np.array_equal(got_n, want_n) and np.array_equal(got_mix, want_mix)
            sel = np.arange(0, len(cnt), 97)
            pairs = np.stack([w1[sel], w0[sel]], axis=1)
            assert np.unique(pairs, axis=0).shape[0] == sel.size         # sampled: a k-mer lives in exactly one task
        del r, w0, w1, cnt, up
    assert sums[0] == sums[1] and sums[0][0] > 250_000_000


def test_full_size_properties_extension(H):
    """Third record shape at full size: the 10 Gbp of reads at K=31 with EXTENSION=1 (8.0e9 k-mers, each carrying PosInRead and
    ReadId through the sort), unfiltered, the result left in HBM and inspected task by task: the counts of a task's entries add up
    to its payload length, payload slices tile the payload array exactly (payload_off = running sum of counts), every position is
    a k-mer start of a 150-base read, every read id is one of this rank's reads, and the totals equal the number of k-mers."""
    G, RL = 312_500_000, 150
    NR = G * 32 // RL
    total = NR * (RL - 31 + 1)
    rid_base = 1000
    from oracle import hsk_oracle as O
    with H.Context(K=31, M=17, L=1, U=65535, EXT=1, ntasks=0, keep_device=True) as c:
        dp, nb, do, dl = c.synth_reads(G, RL, NR, 20251003)
        with H.DeviceDna(c, dp, nb, do, dl, NR).count_resident_device(rid_base=rid_base) as dev:
            # the CPU oracle's digests of every (k-mer, pos, rid) instance, per task: the payload of EVERY entry is checked against them
            want_n, want_mix = _oracle_digests_of_device_reads(c, dp, nb, do, dl, NR, RL, 31, 1, dev.ntasks, rid_base=rid_base)
            seen_pay, seen_cnt, n_entries = 0, 0, 0
            pos_hist = np.zeros(RL - 31 + 1, dtype=np.int64)
            for t in range(dev.ntasks):
                d = dev.fetch(t)
                if not d["n"]:
                    assert want_n[t] == 0
                    continue
                k, cnt = d["kmers"][:, 0], d["cnt"]
                if t == 3 or t == dev.ntasks - 1:                # (expanding 2e8 keys per task on the host takes a while: two of the 40 tasks get the full digest, all of them the counts)
                    rep = np.repeat(d["kmers"], cnt.astype(np.int64), axis=0)
                    with np.errstate(over="ignore"):
                        assert int(O.digest_mix(rep, d["pos"], d["rid"]).sum(dtype=np.uint64)) == int(want_mix[t]), t
                    del rep
                assert int(cnt.sum(dtype=np.uint64)) == int(want_n[t]), t
                assert np.all(k[1:] > k[:-1]) and not np.any(k & np.uint64(3))
                assert int(cnt.sum(dtype=np.uint64)) == d["npay"]
                po = d["payload_off"].astype(np.int64) - d["payload_base"]
                assert po[0] == 0 and np.array_equal(po[1:], np.cumsum(cnt[:-1].astype(np.int64)))
                assert int(d["pos"].max()) <= RL - 31 and int(d["rid"].min()) >= rid_base and int(d["rid"].max()) < rid_base + NR
                pos_hist += np.bincount(d["pos"], minlength=pos_hist.size)
                seen_pay += d["npay"]; seen_cnt += int(cnt.sum(dtype=np.uint64)); n_entries += d["n"]
            assert seen_pay == seen_cnt == total and n_entries == dev.n
            assert np.all(pos_hist == NR)                                 # every k-mer start position of every read exactly once
        c.synth_free(dp, do, dl)


# ---------------------------------------------------------------------------------------------------
# hybrid sort + fused finish (8 tasks per launch): bins with one key, several keys, giant bins
# ---------------------------------------------------------------------------------------------------
def _adversarial_reads(rng, wild):
    g = "".join(rng.choice(list("ACGT"), 30000))
    reads = []
    for _ in range(4000):                                   # ordinary coverage: most bins hold one k-mer, 20x
        p = int(rng.integers(0, len(g) - 150))
        reads.append(g[p:p + 150])
    pre = "ACGTTGCAAGGCTTAACCGG"                            # 20 fixed bases: k-mers starting here share their top 32 bits
    groups = ((60, 3), (700, 2), (160000, 1)) if wild else ((60, 3), (700, 2))
    for nvar, copies in groups:                             # bins with tens / hundreds / thousands of different keys per task
        for v in range(nvar):
            tail = "".join(rng.choice(list("ACGT"), 40))
            reads += [pre + tail] * copies
        pre = pre[1:] + "T"
    reads += ["A" * 400] * 30                               # one k-mer ~11000 times: giant single-key bin
    reads += ["AC" * 150] * 20                              # two k-mers thousands of times each
    reads += [("ACGTACGTAGCTAGCTAGCTAGGATCGATCGATTAGC" * 5)[:150]] * 3000   # short tandem repeat, high counts
    return reads


@pytest.mark.parametrize("L,U,wild", [(1, 65535, True), (2, 50, True), (15, 40, False), (1, 3000, False)])
def test_fused_finish_adversarial(H, O, L, U, wild):
    rng = np.random.default_rng(77)
    seqs = _adversarial_reads(rng, wild)
    dna = H.DnaBuffer.from_sequences(seqs)
    packed, off, lens = dna.arrays()
    for ntasks in (8, 16):
        ores = O.count(packed, off, lens, k=31, m=17, L=L, U=U, ntasks=ntasks, fast=True)
        with H.Context(K=31, M=17, L=L, U=U, ntasks=ntasks) as c:
            res = c.count(dna)
            st = c.stats()
        # the fused kernel and its fallback (a long bin with several keys) must both be exercised over the cases
        assert st["fused_tasks"] + st["redone_tasks"] == ntasks, st
        assert (st["redone_tasks"] > 0) if wild else (st["fused_tasks"] == ntasks), st
        assert np.array_equal(res.task_off, ores.task_off), (L, U, ntasks)
        assert np.array_equal(res.kmers, ores.keys), (L, U, ntasks)
        assert np.array_equal(res.cnt, ores.cnt), (L, U, ntasks)
        assert H.histogram_text(res.histo) == O.histogram_text(ores.cnt)


def test_fused_scatter_long_supermers_one_digit(H, O):
    """Expand fused with the first scatter pass (hsk_scatter.h): supermers of up to 128 k-mers (eight work items each, a
    flush of more than one stage window), a million copies of one k-mer (one digit's chunk list spans hundreds of chunks,
    reservations that straddle three chunks) next to ordinary reads."""
    rng = np.random.default_rng(5)
    g = "".join(rng.choice(list("ACGT"), 40000))
    reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 6000)]
    reads += ["A" * 400] * 3000 + ["ACG" * 120] * 500 + ["T" * 31] * 100
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=31, m=17, L=1, U=65535, ntasks=8, fast=True)
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=8) as c:
        res = c.count(dna)
        st = c.stats()
    assert st["fused_tasks"] + st["redone_tasks"] == 8, st
    assert np.array_equal(res.task_off, ores.task_off)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)


@pytest.mark.parametrize("L,U", [(1, 65535), (2, 50)])
def test_aggregating_finish_table_ladder(H, O, L, U):
    """Prefix bins with ~200 / ~500 / ~1000 distinct keys per task: rank-by-counting, bitonic, and the retry with the
    large hash table (hsk_agg.h) must all give the oracle's list."""
    rng = np.random.default_rng(91)
    g = "".join(rng.choice(list("ACGT"), 20000))
    reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 3000)]
    for pre, nvar in (("AACCGGTTACGTACGGTCAA", 3200), ("ACTGACTGGTCAGTCAACGT", 16000)):
        for v in range(nvar):
            reads.append(pre + "".join(rng.choice(list("ACGT"), 40)))
        reads += reads[-50:]                                   # some of them twice
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    retried = 0
    for ntasks in (8, 16):
        ores = O.count(packed, off, lens, k=31, m=17, L=L, U=U, ntasks=ntasks, fast=True)
        with H.Context(K=31, M=17, L=L, U=U, ntasks=ntasks) as c:
            res = c.count(dna)
            st = c.stats()
        assert st["fused_tasks"] + st["redone_tasks"] == ntasks, st
        retried += st["agg_retried_tasks"]
        assert np.array_equal(res.task_off, ores.task_off), (L, U, ntasks)
        assert np.array_equal(res.kmers, ores.keys), (L, U, ntasks)
        assert np.array_equal(res.cnt, ores.cnt), (L, U, ntasks)
        assert H.histogram_text(res.histo) == O.histogram_text(ores.cnt)
    assert retried > 0


def test_fused_finish_equals_two_pass_path(H):
    """Same input through the aggregating finish (default), the other two plans of the C ABI (HSK_FLAG_NO_AGGREGATION: tile finish,
    HSK_FLAG_FULL_SORT: the reference's algorithm), the single-task path and every test switch of the library (subprocesses: the
    switches are read once)."""
    import subprocess, sys, os, json
    code = ("import os, sys, numpy as np; sys.path.insert(0, %r); import hysortk_amd as H\n"
            "c = H.Context(K=31, M=17, L=2, U=60, ntasks=24, plan=os.environ.get('HSK_TEST_PLAN') or None)\n"
            "dp, nb, do, dl = c.synth_reads(3000000, 150, 400000, 5)\n"
            "r = c.count_device(dp, nb, do, dl, 400000)\n"
            "import hashlib; print(hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes() + r.histo.tobytes()).hexdigest(), len(r))\n") % util.ROOT
    outs = []
    # ... and the parse: fast path (scan/place kernels), the general kernels, and the fast path overflowing its record
    # capacity (falls back to the general kernels; a capacity of 300 is hit by some tiles only)
    for env in ({}, {"HSK_TEST_PLAN": "no_aggregation"}, {"HSK_TEST_PLAN": "full_sort"}, {"HSK_TEST_PLAN": "full_sort", "HSK_XCD_BATCH": "0"}, {"HSK_XCD_BATCH": "0"},
                {"HSK_PARSE_FAST": "0"}, {"HSK_AGG_ADAPT": "2"}, {"HSK_AGG_ADAPT": "2", "HSK_LAG": "1"}, {"HSK_SCAN_GENERIC": "1"}, {"HSK_SCATTER_GENERIC": "1"}, {"HSK_PARSE_REC_CAP": "300"}, {"HSK_PARSE_REC_CAP": "2048"}, {"HSK_WIDE_LOOKBACK": "1"}, {"HSK_WIDE_LOOKBACK": "1", "HSK_XCD_BATCH": "0"}, {"HSK_UNSTABLE_FIRST": "0"}, {"HSK_EXPAND_RESERVE": "0"}, {"HSK_FUSED_SCATTER": "0"}, {"HSK_FORCE_NO_XCD": "1"}, {"HSK_LAG": "0"}, {"HSK_LAG": "1"}, {"HSK_EARLY_D2H": "0"}, {"HSK_COMPACT_D2H": "0"}, {"HSK_COMPACT_D2H": "1"}, {"HSK_WIDEN_THREADS": "3"}, {"HSK_ZERO_COPY": "0"}, {"HSK_XS2": "0"}, {"HSK_PLACE_BYTES": "1"}, {"HSK_DERIVE_OFFSETS": "0"}):
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **util.tune_env(env))).decode().split())
    assert len({o[0] for o in outs}) == 1, outs
    assert int(outs[0][1]) > 100000


@pytest.mark.parametrize("K,EXT", [(31, 1), (51, 0), (77, 0), (51, 1), (31, 0)])
def test_leaving_the_aggregation_in_the_middle_of_a_call(K, EXT):
    """hsk_ctx::agg_off / agg_off_wide (set when a batch finds most bins beyond the tables: input with nearly unique k-mers) change
    the finish of the batches that follow -- tile finish or full-width passes + two-pass counter -- in the middle of a call.
    HSK_AGG_ADAPT=2 sets them after the first batch of three; lists, counts and payload sets must equal the default's and a
    second call on the same context (off from the start) must give the same again.  Subprocesses: the switch is read once."""
    import subprocess, sys, os
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r); import hysortk_amd as H\n"
            "from hysortk_amd import synth\n"
            "seqs = list(synth.reads(300000, 150, 40000, 11)) + ['AC' * 75] * 20\n"
            "dna = H.DnaBuffer.from_sequences(seqs)\n"
            "c = H.Context(K=%d, M=17, L=1, U=65535, EXT=%d, ntasks=24)\n"
            "for it in range(2):\n"
            "    r = c.count(dna, rid_base=5)\n"
            "    h = hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes())\n"
            "    if %d:\n"
            "        own = np.repeat(np.arange(len(r)), r.cnt.astype(np.int64))\n"
            "        sel = np.concatenate([np.arange(int(o), int(o) + int(n)) for o, n in zip(r.payload_off[:-1], r.cnt)])\n"
            "        trip = np.stack([own, r.rid[sel].astype(np.int64), r.pos[sel].astype(np.int64)], 1)\n"
            "        trip = trip[np.lexsort((trip[:, 2], trip[:, 1], trip[:, 0]))]\n"
            "        h.update(trip.tobytes())\n"
            "    print(h.hexdigest(), len(r))\n") % (util.ROOT, K, EXT, EXT)
    outs = []
    for env in ({}, {"HSK_AGG_ADAPT": "2"}, {"HSK_AGG_ADAPT": "2", "HSK_LAG": "1"}):
        outs += [l.split() for l in subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **util.tune_env(env))).decode().strip().splitlines()]
    assert len(outs) == 6 and len({o[0] for o in outs}) == 1, outs
    assert int(outs[0][1]) > 100000


@pytest.mark.parametrize("K,L,U", [(51, 1, 65535), (51, 2, 50), (41, 2, 50), (63, 1, 65535), (35, 2, 50), (33, 1, 65535), (36, 2, 50), (39, 2, 50)])
def test_two_word_keys_prefix_sort_and_aggregation(H, O, K, L, U):
    """32 < K < 64: two scatter passes on the top 16 bits of the most significant word + aggregation of 128-bit keys in LDS
    (slot claimed on word 1, word 0 published by the claimer).  Many k-mers here share word 1 and differ only in word 0
    (variants of one 40-base suffix), others share word 0; K=35, 33, 39 (fewer than 16 prefix bits in word 1): the prefix continues in word 0."""
    rng = np.random.default_rng(K)
    g = "".join(rng.choice(list("ACGT"), 30000))
    reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 4000)]
    suffix = "".join(rng.choice(list("ACGT"), 45))
    prefix = "".join(rng.choice(list("ACGT"), 45))
    for v in range(1500):
        reads.append("".join(rng.choice(list("ACGT"), 40)) + suffix)          # same last bases, different first
        reads.append(prefix + "".join(rng.choice(list("ACGT"), 40)))          # same first bases, different last
    reads += reads[-300:]
    reads += ["AC" * 75] * 40 + ["A" * 150] * 20
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=K, m=17, L=L, U=U, ntasks=16, fast=True)
    with H.Context(K=K, M=17, L=L, U=U, ntasks=16) as c:
        res = c.count(dna)
        st = c.stats()
    assert st["fused_tasks"] + st["redone_tasks"] == 16 and st["fused_tasks"] > 0, st
    assert np.array_equal(res.task_off, ores.task_off)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)
    assert H.histogram_text(res.histo) == O.histogram_text(ores.cnt)


@pytest.mark.parametrize("K,L,U", [(77, 1, 65535), (77, 2, 50), (72, 2, 50), (95, 1, 65535), (69, 2, 50), (65, 1, 65535), (71, 2, 50)])
def test_three_word_keys_prefix_sort_and_aggregation(H, O, K, L, U):
    """64 < K <= 95: two scatter passes on the top 16 bits of the most significant word + aggregation of 192-bit keys in LDS
    (slot claimed on word 2, words 1 and 0 published through the slot's count).  Variants that share the last bases (word 2 and
    the bin) and differ in the first, variants that share the first, k-mers that begin with 32 T and end with 32 A (every value of
    word 0 is a real one), repeats; K=69, 65, 71 (fewer than 16 prefix bits in word 2): the prefix continues in word 1."""
    rng = np.random.default_rng(K)
    g = "".join(rng.choice(list("ACGT"), 30000))
    reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 4000)]
    suffix = "".join(rng.choice(list("ACGT"), 100))
    prefix = "".join(rng.choice(list("ACGT"), 100))
    for v in range(1500):
        reads.append("".join(rng.choice(list("ACGT"), 30)) + suffix)          # same last bases, different first
        reads.append(prefix + "".join(rng.choice(list("ACGT"), 30)))          # same first bases, different last
    reads += reads[-300:]
    for v in range(200):
        reads.append("T" * 32 + "".join(rng.choice(list("ACGT"), K - 64 + 10)) + "A" * 32)
    reads += ["AC" * 75] * 40 + ["A" * 150] * 20
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=K, m=17, L=L, U=U, ntasks=16, fast=True)
    with H.Context(K=K, M=17, L=L, U=U, ntasks=16) as c:
        res = c.count(dna)
        st = c.stats()
    assert st["fused_tasks"] + st["redone_tasks"] == 16 and st["fused_tasks"] > 0, st
    assert np.array_equal(res.task_off, ores.task_off)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)
    assert H.histogram_text(res.histo) == O.histogram_text(ores.cnt)


@pytest.mark.parametrize("L,U", [(1, 65535), (2, 40)])
def test_extension_grouping_aggregation(H, O, L, U):
    """EXTENSION=1 through the batch path (>= 8 tasks): two passes on the top 16 bits with the payload carried, then
    agg_ext_kernel groups every prefix bin by key and writes (pos, rid) to the entry's slice.  Every kept k-mer's payload set
    must equal the oracle's; prefix-sharing variants and repeats go through the table ladder and the long way."""
    from hysortk_amd import synth
    rng = np.random.default_rng(3)
    seqs = list(synth.reads(150000, 150, 9000, 17))
    pre = "ACGTTGCAAGGCTTAACCGG"
    seqs += [pre + "".join(rng.choice(list("ACGT"), 40)) for _ in range(2500)]
    seqs += ["AC" * 75] * 30 + [("ACGGTCATTGCA" * 13)[:150]] * 200
    seqs += ["A" * 150] * 100                      # one bin of 12 000 records: more than the second sweep keeps in registers (8192)
    dna = H.DnaBuffer.from_sequences(seqs)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=31, m=17, L=L, U=U, ext=1, ntasks=16, rid_base=1000, fast=True)
    with H.Context(K=31, M=17, L=L, U=U, EXT=1, ntasks=16) as c:
        res = c.count(dna, rid_base=1000)
        st = c.stats()
    assert st["fused_tasks"] + st["redone_tasks"] == 16 and st["fused_tasks"] > 0, st
    assert np.array_equal(res.task_off, ores.task_off)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)
    assert H.histogram_text(res.histo) == O.histogram_text(ores.cnt)
    for i in list(range(0, len(res), 37)) + [len(res) - 1]:
        pos, rid = res.payload(i)
        a, b = int(ores.payoff[i]), int(ores.payoff[i + 1])
        assert len(pos) == int(res.cnt[i])
        assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(ores.rid[a:b].tolist(), ores.pos[a:b].tolist())), i
    sel = np.concatenate([np.arange(int(o), int(o) + int(c)) for o, c in zip(res.payload_off[:-1], res.cnt)])
    assert sorted(zip(res.rid[sel].tolist(), res.pos[sel].tolist())) == sorted(zip(ores.rid.tolist(), ores.pos.tolist()))


@pytest.mark.parametrize("K,L,U", [(51, 1, 65535), (51, 2, 40), (77, 2, 40), (35, 1, 65535), (69, 2, 40)])
def test_extension_multiword_keys_grouping_aggregation(H, O, K, L, U):
    """EXTENSION=1 with keys of two and three words through the batch path: prefix passes with the payload carried (the prefix
    continues in the word below for K=35 / 69), then agg_ext_kernel<cap, NW> groups every prefix bin by key.  Entries, counts and every
    kept k-mer's payload set against the oracle; prefix-sharing variants go up the table ladder, a poly-A bin exceeds the
    second sweep's registers."""
    from hysortk_amd import synth
    rng = np.random.default_rng(K)
    seqs = list(synth.reads(120000, 150, 6000, 17))
    pre = "".join(rng.choice(list("ACGT"), 100))
    seqs += ["".join(rng.choice(list("ACGT"), 40)) + pre for _ in range(1800)]      # same last bases (the prefix bin), different first
    seqs += ["AC" * 75] * 30 + [("ACGGTCATTGCA" * 13)[:150]] * 150
    seqs += ["A" * 150] * 120
    dna = H.DnaBuffer.from_sequences(seqs)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=K, m=17, L=L, U=U, ext=1, ntasks=16, rid_base=7, fast=True)
    with H.Context(K=K, M=17, L=L, U=U, EXT=1, ntasks=16) as c:
        res = c.count(dna, rid_base=7)
        st = c.stats()
    assert st["fused_tasks"] + st["redone_tasks"] == 16 and st["fused_tasks"] > 0, st
    assert np.array_equal(res.task_off, ores.task_off)
    assert np.array_equal(res.kmers, ores.keys)
    assert np.array_equal(res.cnt, ores.cnt)
    assert H.histogram_text(res.histo) == O.histogram_text(ores.cnt)
    for i in list(range(0, len(res), 41)) + [len(res) - 1]:
        pos, rid = res.payload(i)
        a, b = int(ores.payoff[i]), int(ores.payoff[i + 1])
        assert len(pos) == int(res.cnt[i])
        assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(ores.rid[a:b].tolist(), ores.pos[a:b].tolist())), i
    sel = np.concatenate([np.arange(int(o), int(o) + int(c)) for o, c in zip(res.payload_off[:-1], res.cnt)])
    assert sorted(zip(res.rid[sel].tolist(), res.pos[sel].tolist())) == sorted(zip(ores.rid.tolist(), ores.pos.tolist()))


def test_resident_result_csr_on_device(H, O):
    """HSK_FLAG_KEEP_DEVICE: the list stays in HBM and hsk_result_device_task hands out the per-task entry arrays and, with
    EXTENSION, the CSR payload (payload_off / pos / rid) for a following GPU stage; read back they equal the host result."""
    from hysortk_amd import synth
    seqs = synth.reads(120000, 150, 8000, 29)
    dna = H.DnaBuffer.from_sequences(seqs)
    with H.Context(K=31, M=17, L=2, U=50, EXT=1, ntasks=16) as c:
        host = c.count(dna, rid_base=7)
    with H.Context(K=31, M=17, L=2, U=50, EXT=1, ntasks=16, keep_device=True) as c:
        with c.count_resident(dna, rid_base=7) as dev:
            assert dev.n == len(host) and np.array_equal(dev.task_off, host.task_off)
            got = 0
            for t in range(dev.ntasks):
                a, b = int(host.task_off[t]), int(host.task_off[t + 1])
                d = dev.fetch(t)
                assert d["n"] == b - a
                if not d["n"]:
                    continue
                assert np.array_equal(d["kmers"], host.kmers[a:b]) and np.array_equal(d["cnt"], host.cnt[a:b])
                assert np.array_equal(d["payload_off"], host.payload_off[a:b])      # same numbering as the host arrays
                for i in range(0, d["n"], 41):
                    first = int(d["payload_off"][i]) - d["payload_base"]
                    cnt = int(d["cnt"][i])
                    hp, hr = host.payload(a + i)
                    assert sorted(zip(d["rid"][first:first + cnt].tolist(), d["pos"][first:first + cnt].tolist())) == sorted(zip(hr.tolist(), hp.tolist()))
                got += d["n"]
            assert got == len(host)


def test_fasta_ingest_on_device(H, O, tmp_path):
    """hsk_pack_fasta: FASTA text -> DnaBuffer bytes in HBM must equal the host packer byte for byte (line breaks at any width,
    CRLF, lower case, N, a character outside ACGTN with the reference's code-4 spill, empty and 1-base records), and counting
    from the device buffer equals counting from the host DnaBuffer."""
    rng = np.random.default_rng(11)
    seqs = ["".join(rng.choice(list("ACGT"), n)) for n in (3000, 1500, 10, 31, 250, 1, 0, 77, 4, 5, 801)]
    seqs[3] = seqs[3].lower()
    seqs[4] = seqs[4][:100] + "NNNNnnnn" + seqs[4][108:]
    seqs[7] = seqs[7][:30] + "R" + seqs[7][31:]                     # non-nucleotide: code 4 corrupts the neighbouring bits as in the reference
    fa = tmp_path / "x.fa"
    with open(fa, "wb") as f, open(str(fa) + ".fai", "w") as fai:
        for i, s in enumerate(seqs):
            width, eol = ((60, b"\n"), (80, b"\r\n"), (7, b"\n"))[i % 3]
            f.write(b">r%d some text\n" % i)
            pos = f.tell()
            for j in range(0, len(s), width):
                f.write(s[j:j + width].encode() + eol)
            if not s:
                f.write(b"\n")
            fai.write("r%d\t%d\t%d\t%d\t%d\n" % (i, len(s), pos, width, width + len(eol)))
    host = H.DnaBuffer.from_sequences(seqs)
    packed, off, lens = host.arrays()
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=3) as c:
        dd = H.read_dna_buffer_device(c, str(fa))
        assert dd.nreads == len(seqs) and dd.nbytes == packed.size
        assert dd.packed().tobytes() == packed.tobytes()
        res_d = dd.count()
        dd.free()
        res_h = c.count(host)
    assert np.array_equal(res_d.kmers, res_h.kmers) and np.array_equal(res_d.cnt, res_h.cnt) and np.array_equal(res_d.task_off, res_h.task_off)
    # and the golden FASTA of the reference runs: same list as through read_dna_buffer
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=5) as c:
        dd = H.read_dna_buffer_device(c, util.GOLDEN + "/reads_small.fa")
        res_d = dd.count()
        dd.free()
        res_h = c.count(H.read_dna_buffer(util.GOLDEN + "/reads_small.fa"))
    assert np.array_equal(res_d.kmers, res_h.kmers) and np.array_equal(res_d.cnt, res_h.cnt)


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_configurations_vs_oracle(H, O, seed):
    """Seeded sweep over K (one, two and three key words), M (window widths below and above 8, M > 25 takes the general parse
    kernels), task counts (single-task path, padded batches, full batches), L/U, EXTENSION and ragged reads (shorter than K,
    exactly K, with N and lower case): every combination must give the oracle's list."""
    rng = np.random.default_rng(1000 + seed)
    K = int(rng.choice([5, 11, 15, 21, 27, 31, 33, 39, 41, 51, 63, 65, 77, 95]))
    M = int(rng.integers(max(1, min(K - 60, 20)), min(K, 32)))
    if seed % 4 == 3 and K > 34:                                  # multi-word minimizers (M > 32; M % 32 == 0 is rejected like K % 32 == 0)
        M = int(rng.choice([m for m in range(33, K) if m % 32]))
    EXT = int(rng.integers(0, 2)) if K < 64 else 0
    ntasks = int(rng.choice([1, 2, 5, 8, 11, 16, 24]))
    L = int(rng.choice([1, 1, 2, 3])); U = int(rng.choice([4, 40, 65535]))
    if U < L:
        U = L
    g = "".join(rng.choice(list("ACGT"), 40000))
    reads = []
    for _ in range(1500):
        n = int(rng.choice([rng.integers(1, K + 2), K, rng.integers(K, 400), 150]))
        p = int(rng.integers(0, len(g) - n))
        s = g[p:p + n]
        if rng.random() < 0.5:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        if rng.random() < 0.05 and n > 3:
            q = int(rng.integers(0, n - 1)); s = s[:q] + "N" + s[q + 1:]
        if rng.random() < 0.05:
            s = s.lower()
        reads.append(s)
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=K, m=M, L=L, U=U, ext=EXT, ntasks=ntasks, rid_base=seed, fast=True)
    with H.Context(K=K, M=M, L=L, U=U, EXT=EXT, ntasks=ntasks) as c:
        res = c.count(dna, rid_base=seed)
    tag = (K, M, EXT, ntasks, L, U)
    assert np.array_equal(res.task_off, ores.task_off), tag
    assert np.array_equal(res.kmers, ores.keys), tag
    assert np.array_equal(res.cnt, ores.cnt), tag
    assert H.histogram_text(res.histo) == O.histogram_text(ores.cnt), tag
    if EXT and len(res):
        for i in list(range(0, len(res), max(1, len(res) // 40))) + [len(res) - 1]:
            pos, rid = res.payload(i)
            a, b = int(ores.payoff[i]), int(ores.payoff[i + 1])
            assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(ores.rid[a:b].tolist(), ores.pos[a:b].tolist())), (tag, i)


def test_extension_fused_scatter_equals_two_pass_path():
    """EXTENSION through the expand fused with the first scatter pass (payload chunks beside the key chunks, default) and
    through expand + two passes with payload (HSK_FUSED_SCATTER_EXT=0): same k-mers, counts, task offsets, and the same
    payload multiset per k-mer (order inside one k-mer is free: an order-independent sum of hashed (pos, rid) pairs)."""
    import subprocess, sys, os
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import hysortk_amd as H\n"
            "c = H.Context(K=31, M=17, L=2, U=60, EXT=1, ntasks=16)\n"
            "dp, nb, do, dl = c.synth_reads(3000000, 150, 400000, 5)\n"
            "r = c.count_device(dp, nb, do, dl, 400000, rid_base=7)\n"
            "h = (r.pos.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (r.rid.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F))\n"
            "per = np.add.reduceat(h, r.payload_off[:len(r)].astype(np.int64)) if len(r) else h[:0]\n"
            "import hashlib; print(hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes() + per.tobytes()).hexdigest(), len(r), len(r.pos))\n") % util.ROOT
    outs = [subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **util.tune_env(env))).decode().split()
            for env in ({}, {"HSK_FUSED_SCATTER_EXT": "0"}, {"HSK_FUSED_SCATTER": "0"})]
    assert outs[0] == outs[1] == outs[2], outs
    assert int(outs[0][1]) > 100000 and int(outs[0][2]) > int(outs[0][1])


@pytest.mark.parametrize("K", [51, 41, 63])
def test_two_word_keys_fused_scatter_equals_two_pass_path(K):
    """32 < K < 64 through the fused expand + scatter (2048-key chunks of 16-byte keys, items of 8 k-mers; default) and through
    expand + two passes (HSK_FUSED_SCATTER_WIDE=0): byte-identical lists."""
    import subprocess, sys, os
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import hysortk_amd as H\n"
            "c = H.Context(K=%d, M=17, L=2, U=60, ntasks=16)\n"
            "dp, nb, do, dl = c.synth_reads(3000000, 150, 400000, 5)\n"
            "r = c.count_device(dp, nb, do, dl, 400000)\n"
            "import hashlib; print(hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes() + r.histo.tobytes()).hexdigest(), len(r), c.stats()['fused_tasks'])\n") % (util.ROOT, K)
    outs = [subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **util.tune_env(env))).decode().split()
            for env in ({}, {"HSK_FUSED_SCATTER_WIDE": "0"})]
    assert outs[0] == outs[1], outs
    assert int(outs[0][1]) > 100000 and int(outs[0][2]) == 16


@pytest.mark.parametrize("seed", range(6))
def test_fused_scatter_sweep_vs_oracle(H, O, seed):
    """The benchmark path (one-word keys, no payload, whole batches of 8 tasks: expand fused with the first scatter pass,
    chunk-listed bins, second pass over chunk tiles, aggregation) over K, M, task counts and skewed inputs at a few
    million k-mers: digits with thousands of chunks next to empty ones, partial batches padded with empty tasks."""
    rng = np.random.default_rng(4000 + seed)
    K = int(rng.choice([17, 21, 25, 27, 29, 31]))
    M = int(rng.integers(7, min(K - 1, 24)))
    ntasks = int(rng.choice([8, 11, 16, 24, 40]))
    L = int(rng.choice([1, 2, 3])); U = int(rng.choice([40, 65535]))
    g = "".join(rng.choice(list("ACGT"), 200000, p=[0.4, 0.1, 0.1, 0.4] if seed % 2 else [0.25] * 4))
    reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 30000)]
    reads += ["".join(rng.choice(list("AT"), 200)) for _ in range(2000)]          # low complexity: few digits, long chunk lists
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=K, m=M, L=L, U=U, ntasks=ntasks, fast=True)
    with H.Context(K=K, M=M, L=L, U=U, ntasks=ntasks) as c:
        res = c.count(dna)
        st = c.stats()
    tag = (K, M, ntasks, L, U)
    assert st["fused_tasks"] + st["redone_tasks"] == ntasks, (tag, st)
    assert np.array_equal(res.task_off, ores.task_off), tag
    assert np.array_equal(res.kmers, ores.keys), tag
    assert np.array_equal(res.cnt, ores.cnt), tag


@pytest.mark.parametrize("seed", range(12))
def test_wide_keys_and_extension_sweep_vs_oracle(H, O, seed):
    """Random (K, M, EXTENSION, task count, filter) over keys of one to three words, every length of the most significant word
    (the prefix inside it or continued in the word below), whole batches of 8 tasks: prefix passes + agg2 / agg3 /
    agg_ext kernels (one to three key words) against the oracle, entries, counts and (EXTENSION) the payload sets."""
    rng = np.random.default_rng(7000 + seed)
    K = int(rng.choice([33, 34, 37, 39, 40, 47, 55, 63, 65, 66, 70, 71, 72, 80, 93, 95, 31, 27]))
    M = int(rng.integers(9, 24))
    EXT = int(seed % 3 != 0)
    ntasks = int(rng.choice([8, 13, 16, 24]))
    L = int(rng.choice([1, 2])); U = int(rng.choice([40, 65535]))
    g = "".join(rng.choice(list("ACGT"), 60000, p=[0.35, 0.15, 0.15, 0.35] if seed % 2 else [0.25] * 4))
    reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 9000)]
    reads += ["".join(rng.choice(list("AT"), 170)) for _ in range(400)]
    reads += ["A" * 150] * 70 + ["".join(rng.choice(list("ACGT"), 30)) + g[1000:1120] for _ in range(700)]
    dna = H.DnaBuffer.from_sequences(reads)
    packed, off, lens = dna.arrays()
    ores = O.count(packed, off, lens, k=K, m=M, L=L, U=U, ext=EXT, ntasks=ntasks, rid_base=3, fast=True)
    with H.Context(K=K, M=M, L=L, U=U, EXT=EXT, ntasks=ntasks) as c:
        res = c.count(dna, rid_base=3)
        st = c.stats()
    tag = (K, M, EXT, ntasks, L, U)
    assert st["fused_tasks"] + st["redone_tasks"] == ntasks and st["fused_tasks"] > 0, (tag, st)
    assert np.array_equal(res.task_off, ores.task_off), tag
    assert np.array_equal(res.kmers, ores.keys), tag
    assert np.array_equal(res.cnt, ores.cnt), tag
    if EXT:
        for i in list(range(0, len(res), 211)) + [len(res) - 1]:
            pos, rid = res.payload(i)
            a, b = int(ores.payoff[i]), int(ores.payoff[i + 1])
            assert sorted(zip(rid.tolist(), pos.tolist())) == sorted(zip(ores.rid[a:b].tolist(), ores.pos[a:b].tolist())), (tag, i)


@pytest.mark.parametrize("K,M,EXT", [(5, 3, 0), (7, 4, 0), (9, 5, 1), (11, 7, 0), (13, 6, 1), (15, 9, 0), (16, 11, 0), (19, 17, 1)])
def test_short_kmers_vs_oracle(H, O, K, M, EXT):
    """The low end of K (down to keys of 10 bits: fewer than the 16 prefix bits, nearly every prefix bin empty, every k-mer
    thousands of times) with both task counts (single tasks, whole batches)."""
    from hysortk_amd import synth
    seqs = list(synth.reads(40000, 150, 3000, 23)) + ["ACGT" * 30] * 50 + ["A" * 150] * 10
    dna = H.DnaBuffer.from_sequences(seqs)
    packed, off, lens = dna.arrays()
    for ntasks in (3, 16):
        ores = O.count(packed, off, lens, k=K, m=M, L=1, U=65535, ext=EXT, ntasks=ntasks, rid_base=2, fast=True)
        with H.Context(K=K, M=M, L=1, U=65535, EXT=EXT, ntasks=ntasks) as c:
            res = c.count(dna, rid_base=2)
        tag = (K, M, EXT, ntasks)
        assert np.array_equal(res.task_off, ores.task_off), tag
        assert np.array_equal(res.kmers, ores.keys), tag
        assert np.array_equal(res.cnt, ores.cnt), tag
        if EXT:
            sel = np.concatenate([np.arange(int(o), int(o) + int(n)) for o, n in zip(res.payload_off[:-1], res.cnt)])
            assert sorted(zip(res.rid[sel].tolist(), res.pos[sel].tolist())) == sorted(zip(ores.rid.tolist(), ores.pos.tolist())), tag


@pytest.mark.parametrize("K", [31, 51, 77])
def test_output_text_formatted_on_device(H, O, K, tmp_path):
    """hsk_format_entries: the "KMER\\tcount" lines of write_output_file (reference src/hysortk.cpp:138-164) formatted on the GPU
    equal the host formatter byte for byte (one, two and three key words; counts of one to five digits)."""
    from hysortk_amd import synth
    seqs = list(synth.reads(60000, 150, 3000, 41)) + ["ACGT" * 40] * 1200 + [("ACGGTCATTGCA" * 13)[:150]] * 2000
    dna = H.DnaBuffer.from_sequences(seqs)
    with H.Context(K=K, M=17, L=1, U=65535, ntasks=5) as c:
        kl = c.count(dna)
        assert int(kl.cnt.max()) >= 10000 and int(kl.cnt.min()) == 1
        (tmp_path / "g").mkdir(); (tmp_path / "h").mkdir()
        H.write_output_file(kl, str(tmp_path / "g"), ctx=c)
        H.write_output_file(kl, str(tmp_path / "h"))
        assert c.format_entries(kl.kmers[:0], kl.cnt[:0]) == b""
    g = open(tmp_path / "g" / "0.out", "rb").read()
    h = open(tmp_path / "h" / "0.out", "rb").read()
    assert g == h and g.count(b"\n") == len(kl)


def test_eighty_gbp_on_one_gpu(H):
    """The whole input of BASELINE configs[2] (80 Gbp: 6.4e10 31-mers, 304 tasks) in ONE call on one GPU, unfiltered, the 2.5e9-entry list
    left in HBM (205 GB of the 288 GB live at the peak).  Size-independent properties: total_kmers, the count histogram adds up to the number
    of k-mers and to the number of entries, the entries are the genome's distinct positions (bar a handful of chance repeats), and a second
    call gives the same histogram."""
    G, RL = 2_500_000_000, 150
    NR = G * 32 // RL
    total = NR * (RL - 31 + 1)
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=0, keep_device=True) as c:
        try:
            dp, nb, do, dl = c.synth_reads(G, RL, NR, 424242)
        except H.HskError as e:
            pytest.skip("no room for 20 GB of synthetic reads: %s" % e)
        hs = []
        for _ in range(2):
            r = c.count_device(dp, nb, do, dl, NR)
            h = np.asarray(r.histo, dtype=np.uint64)
            assert int(r.info["total_kmers"]) == total
            assert int((h * np.arange(h.size, dtype=np.uint64)).sum(dtype=np.uint64)) == total and int(h.sum(dtype=np.uint64)) == int(r.info["n"])
            assert abs(int(r.info["n"]) - (G - 31 + 1)) < 1000
            hs.append(h.copy())
            del r
        assert np.array_equal(hs[0], hs[1])
        c.synth_free(dp, do, dl)
