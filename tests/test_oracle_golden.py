"""Pins the CPU restatement (oracle/hsk_oracle.c) against outputs of the REAL reference
(tests/golden/, produced by tests/golden/make_golden.py from /root/reference).  CPU only."""
import numpy as np
import pytest

from oracle import hsk_oracle as O
from tests import util


def test_murmur_known_answers():
    g = util.load_json("murmur.json")["murmur"]
    assert len(g) >= 60
    for e in g:
        key = util.hex_words(e["key"])
        assert O.murmur64(key) == int(e["hash"], 16), e
    # SURVEY 8a KATs
    assert O.murmur64([0]) == 0x864BA144DF098483
    assert O.murmur64([0x1BE429F040000000]) == 0x465BB05EB02CF5F5


@pytest.mark.parametrize("variant", ["k31", "k31ext", "k51", "k21", "k51m35", "k77m65"])
def test_stage_vectors(variant):
    cfg = util.VARIANTS[variant]
    g = util.load_json("stages_%s.json" % variant)
    k, m = g["K"], g["M"]
    assert (k, m) == (cfg["k"], cfg["m"])
    for rd in g["reads"]:
        seq = rd["seq"]
        packed = O.pack(seq)
        assert packed.tobytes().hex() == rd["packed"]                       # a1 DnaSeq::compress
        rep = O.rep_mers(packed, len(seq), k)                               # a2 GetRepKmers
        assert rep.shape[0] == len(rd["repkmers"])
        if rd["repkmers"]:
            want = np.array([[int(x, 16) for x in ws] for ws in rd["repkmers"]], dtype=np.uint64)
            assert np.array_equal(rep, want)
        hashes = O.mmer_hashes(packed, len(seq), m)                         # a3 GetRepMmers + GetHash
        assert np.array_equal(hashes, util.hex_words(rd["mmerhash"]))
        for tot, tv in rd["tasks"].items():
            d = O.dests(packed, len(seq), k, m, int(tot))                   # a4
            assert d.tolist() == tv["dest"]
            sm = O.supermers(d, k, packed)                                  # a5
            # the harness lists supermers grouped by task (ascending), in read order inside a task
            sm_sorted = sorted(range(len(sm)), key=lambda i: (sm[i][0], i))
            got = [dict(task=sm[i][0], len=sm[i][2], bytes=sm[i][3].tobytes().hex()) for i in sm_sorted]
            wantsm = [dict(task=x["task"], len=x["len"], bytes=x["bytes"]) for x in tv["supermers"]]
            assert got == wantsm
            if g["EXT"]:
                assert [sm[i][1] for i in sm_sorted] == [x["pos"] for x in tv["supermers"]]


def _oracle_count(variant, ntasks=5, **kw):
    cfg = dict(util.VARIANTS[variant])
    cfg.update(kw)
    seqs = util.read_fasta(util.GOLDEN + "/reads_small.fa")
    packed, off, lens = O.pack_reads(seqs)
    return cfg, O.count(packed, off, lens, ntasks=ntasks, **cfg)


@pytest.mark.parametrize("variant", ["k31", "k31f", "k21", "k51", "k51m35", "k77m65"])
def test_count_raw_order(variant):
    """1 rank x 8 threads -> tot_tasks = 5; the raw vector (per-task ascending runs, ascending task id)
    is reproduced element for element (k51: RADULS order = little-endian multiword)."""
    cfg, res = _oracle_count(variant)
    gold = util.load_count("count_%s.txt" % variant)
    got = util.result_strings(res.keys, cfg["k"])
    assert len(got) == len(gold)
    assert got == [g[0] for g in gold]
    assert res.cnt.tolist() == [g[1] for g in gold]
    assert O.histogram_text(res.cnt) == open(util.GOLDEN + "/hist_%s.txt" % variant).read()


def test_count_k51_paradis_multiset():
    """PARADIS (hybrid MSD + std::sort) orders K>32 differently inside a task; content is identical."""
    cfg, res = _oracle_count("k51p")
    gold = util.load_count("count_k51p.txt")
    got = sorted(zip(util.result_strings(res.keys, cfg["k"]), res.cnt.tolist()))
    assert got == sorted((g[0], g[1]) for g in gold)
    assert O.histogram_text(res.cnt) == open(util.GOLDEN + "/hist_k51p.txt").read()


def test_count_extension_payload():
    cfg, res = _oracle_count("k31ext")
    gold = util.load_count("count_k31ext.txt")
    got = util.result_strings(res.keys, cfg["k"])
    assert got == [g[0] for g in gold]
    assert res.cnt.tolist() == [g[1] for g in gold]
    # payload order inside a k-mer is unspecified in the reference (unstable sorts): compare as sets
    for i, g in enumerate(gold):
        a, b = int(res.payoff[i]), int(res.payoff[i + 1])
        assert sorted(zip(res.rid[a:b].tolist(), res.pos[a:b].tolist())) == sorted(zip(g[3], g[2])), g[0]


@pytest.mark.parametrize("nprocs", [2, 3])
def test_multirank_union_and_dispatch(nprocs):
    """mpiexec -n {2,3} x 4 threads of the reference: sorted union of the rank outputs equals the
    oracle with tot_tasks = hsko_tot_tasks(4, nprocs); the dispatcher restatement reproduces the
    reference's task -> rank table from the LOG=2 task sizes."""
    d = util.load_json("dispatch_k31_np%d.json" % nprocs)
    tot = O.tot_tasks(d["omp_threads"], nprocs)
    assert tot == len(d["task_bytes"])
    owner = O.dispatch_balanced(d["task_bytes"], nprocs)
    table = {int(r): ids for r, ids in d["task_ids_per_rank"].items()}
    for r, ids in table.items():
        assert [t for t in range(tot) if owner[t] == r] == ids
    assert O.classify([1] * tot).tolist() == [0] * tot
    cfg, res = _oracle_count("k31", ntasks=tot)
    lines = sorted("%s\t%d" % (s, c) for s, c in zip(util.result_strings(res.keys, 31), res.cnt.tolist()))
    gold = open(util.GOLDEN + "/count_k31_np%d.txt" % nprocs).read().splitlines()
    if d["task_types"] and any(d["task_types"]):
        pytest.skip("heavy-hitter path taken by the reference for this run")
    assert lines == gold
    # per-rank entry counts = sum over the rank's tasks
    for r, ids in table.items():
        n = sum(int(res.task_off[t + 1] - res.task_off[t]) for t in ids)
        assert n == d["entries_per_rank"][r]
    assert O.histogram_text(res.cnt) == d["histogram"]


@pytest.mark.parametrize("k,m,ext,nt", [(31, 17, 0, 5), (31, 17, 1, 7), (51, 17, 0, 40), (51, 35, 0, 5), (77, 65, 0, 6), (21, 9, 0, 3), (77, 17, 1, 9), (35, 17, 0, 8)])
def test_streaming_task_digests_equal_the_counted_lists(k, m, ext, nt):
    """hsko_task_digests (rolling, streaming: the checker of the full-size GPU tests, where hsko_count cannot hold the k-mers) against
    the digests of hsko_count's own lists -- itself pinned by the goldens above -- on the golden reads plus edge cases (reads
    shorter than / equal to K, homopolymers, tandem repeats, N / lower case in the golden file)."""
    seqs = util.read_fasta(util.GOLDEN + "/reads_small.fa") + ["ACGT" * 3, "A" * 200, "ACGTTGCA" * 30, "C" * (k - 1), "G" * k, ""]
    pk, off, ln = O.pack_reads(seqs)
    r = O.count(pk, off, ln, k=k, m=m, L=1, U=65535, ext=ext, ntasks=nt, rid_base=11)
    n1, m1 = O.entries_digests(r.keys, r.cnt, r.task_off, (r.payoff, r.pos, r.rid) if ext else None)
    n2, m2 = O.task_digests(pk, off, ln, k=k, m=m, ext=ext, ntasks=nt, rid_base=11)
    assert np.array_equal(n1, n2) and np.array_equal(m1, m2)
    assert int(n2.sum()) == r.stats["total_kmers"]
    sel = np.zeros(nt, np.uint8); sel[nt // 2] = 1
    n3, m3 = O.task_digests(pk, off, ln, k=k, m=m, ext=ext, ntasks=nt, rid_base=11, task_sel=sel)
    assert n3[nt // 2] == n2[nt // 2] and m3[nt // 2] == m2[nt // 2] and int(n3.sum()) == int(n2[nt // 2])
