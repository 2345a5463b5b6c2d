"""Host-side logic of the product (no GPU): packing, DnaBuffer, planning rules, FASTA ingest,
histogram text, synthetic generator -- against the oracle and the reference's golden vectors."""
import io
import os

import numpy as np
import pytest

import hysortk_amd as H
from hysortk_amd import synth
from oracle import hsk_oracle as O
from tests import util


def test_pack_sequence_equals_reference_bytes_and_oracle():
    for v in ("k31", "k51"):
        for rd in util.load_json("stages_%s.json" % v)["reads"]:
            assert H.pack_sequence(rd["seq"]).tobytes().hex() == rd["packed"]
    rng = np.random.default_rng(0)
    for n in (0, 1, 3, 4, 5, 63, 64, 65, 1000):
        s = "".join(rng.choice(list("ACGTacgtNnXx-"), n))
        assert np.array_equal(H.pack_sequence(s), O.pack(s)), (n, s[:20])


def test_dnabuffer_layout():
    seqs = ["ACGTA", "", "TTTTGGGGC", "A" * 40]
    b = H.DnaBuffer.from_sequences(seqs)
    packed, off, lens = b.arrays()
    assert b.size() == 4 and lens.tolist() == [5, 0, 9, 40]
    assert off.tolist() == [0, 2, 2, 5] and b.getbufsize() == 15          # every read starts on a byte boundary
    assert [b[i].ascii() for i in range(4)] == seqs
    assert b[2][4] == 2 and b[2].numbytes() == 3
    op, oo, ol = O.pack_reads(seqs)
    assert np.array_equal(packed, op) and np.array_equal(off, oo) and np.array_equal(lens, ol)


def test_plan_tot_tasks_rule():
    for thr in (1, 2, 4, 7, 8, 16, 64, 256):
        for nprocs in (1, 2, 8):
            assert H.plan_tot_tasks(thr, nprocs) == O.tot_tasks(thr, nprocs)
    assert H.plan_tot_tasks(8, 1) == 5 and H.plan_tot_tasks(2, 3) == 9


@pytest.mark.parametrize("nprocs", [2, 3])
def test_dispatch_reproduces_reference_table(nprocs):
    d = util.load_json("dispatch_k31_np%d.json" % nprocs)
    owner = H.plan_dispatch(d["task_bytes"], nprocs)
    for r, ids in d["task_ids_per_rank"].items():
        assert [t for t in range(len(owner)) if owner[t] == int(r)] == ids
    assert np.array_equal(owner, O.dispatch_balanced(d["task_bytes"], nprocs))


def test_dispatch_random_vs_oracle_and_errors():
    rng = np.random.default_rng(5)
    for _ in range(200):
        nprocs = int(rng.integers(1, 9))
        ntasks = int(rng.integers(nprocs, 60))
        sz = np.unique(rng.integers(1000, 100000, size=ntasks * 2))[:ntasks]   # distinct sizes: std::sort ties are unspecified
        rng.shuffle(sz)
        if sz.size < nprocs:
            continue
        try:
            want = O.dispatch_balanced(sz, nprocs)
        except RuntimeError:
            with pytest.raises(H.HskError):
                H.plan_dispatch(sz, nprocs)
            continue
        assert np.array_equal(H.plan_dispatch(sz, nprocs), want)
    # the reference throws "Cannot dispatch tasks" on hopeless skew
    skew = [100, 100, 100, 100, 100]                 # 4 ranks: the fifth task would need a cap of 1.6 x average
    with pytest.raises(RuntimeError):
        O.dispatch_balanced(skew, 4)
    with pytest.raises(H.HskError) as e:
        H.plan_dispatch(skew, 4)
    assert "Cannot dispatch" in str(e.value)
    assert H.plan_dispatch([5, 6, 7, 8, 9], 2, plain=True).tolist() == [0, 1, 0, 1, 0]


def test_classify_rule():
    k = [100, 100, 100, 1000, 100, 100]
    assert np.array_equal(H.plan_classify(k), O.classify(k))
    assert H.plan_classify(k).tolist() == [0, 0, 0, 1, 0, 0]


def test_partition_reads_rule():
    lens = [150] * 100
    assert H.plan_partition_reads(lens, 4).tolist() == [24, 24, 24, 28]     # the last rank takes the remainder (fastaindex.cpp:95-99)
    assert H.plan_partition_reads([1000, 10, 10, 10], 2).tolist() == [1, 3]
    assert int(H.plan_partition_reads(np.arange(1, 1000), 7).sum()) == 999


def test_read_dna_buffer_and_fai(tmp_path):
    seqs = util.read_fasta(util.GOLDEN + "/reads_small.fa")
    recs = H.read_fai(util.GOLDEN + "/reads_small.fa.fai")
    assert [r[1] for r in recs] == [len(s) for s in seqs]
    dna = H.read_dna_buffer(util.GOLDEN + "/reads_small.fa")
    assert dna.size() == len(seqs)
    ref = H.DnaBuffer.from_sequences(seqs)
    for a, b in zip(dna.arrays(), ref.arrays()):
        assert np.array_equal(a, b)


def test_histogram_text_format():
    gold = open(util.GOLDEN + "/hist_k31.txt").read()
    cnt = np.array([g[1] for g in util.load_count("count_k31.txt")], dtype=np.uint64)
    histo = np.bincount(cnt.astype(np.int64), minlength=int(cnt.max()) + 1)
    assert H.histogram_text(histo) == gold == O.histogram_text(cnt)
    kl = H.KmerList(31, np.zeros((0, 1), np.uint64), np.zeros(0, np.uint64), np.zeros(2, np.uint64), histo=histo)
    buf = io.StringIO()
    H.print_kmer_histogram(kl, file=buf)
    assert buf.getvalue() == gold


def test_synth_twin_is_deterministic_and_consistent():
    a = synth.reads(5000, 150, 50, 9)
    b = synth.reads(5000, 150, 50, 9)
    assert a == b and len(set(a)) > 40 and all(len(s) == 150 for s in a)
    packed, off, lens = synth.packed_reads(5000, 150, 50, 9)
    op, oo, ol = O.pack_reads(a)
    assert np.array_equal(packed, op) and np.array_equal(off, oo) and np.array_equal(lens, ol)
    # first_read selects a window of the same global read stream
    assert synth.reads(5000, 150, 10, 9, first_read=20) == a[20:30]


def test_plan_exchange_covers_everything_exactly():
    rng = np.random.default_rng(3)
    nranks, ntasks = 4, 11
    owner = rng.integers(0, nranks, size=ntasks).astype(np.int32)
    M = rng.integers(0, 50, size=(nranks, ntasks, 3)).astype(np.uint64)
    M[:, :, 1] = M[:, :, 0] * 9
    M[:, :, 2] = M[:, :, 0] * 8
    plans = [H.plan_exchange(nranks, r, owner, M) for r in range(nranks)]
    for r in range(nranks):
        sr, segs = plans[r]
        for q in range(nranks):                       # what r sends to q is what q receives from r
            assert sr[q, 0] == plans[q][0][r, 4] and sr[q, 1] == plans[q][0][r, 5]
            assert sr[q, 0] == M[r, owner == q, 0].sum()
        # receive regions are disjoint and contiguous
        assert sr[:, 6].tolist() == np.concatenate([[0], np.cumsum(sr[:, 4])[:-1]]).tolist()
        for t in range(ntasks):
            if owner[t] != r:
                assert not segs[t].any()
                continue
            assert segs[t, :, 1].tolist() == M[:, t, 0].tolist()
            assert segs[t, :, 3].tolist() == np.concatenate([[0], np.cumsum(M[:, t, 2])[:-1]]).tolist()
            for p in range(nranks):                   # segment lies inside the region received from p
                if segs[t, p, 1]:
                    assert sr[p, 6] <= segs[t, p, 0] and segs[t, p, 0] + segs[t, p, 1] <= sr[p, 6] + sr[p, 4]


def test_paradis_order_rebuilds_the_reference_raw_vector_k51():
    """K=51 built with PARADIS (golden count_k51p.txt, raw KmerListS order of the real reference): the oracle's list in this library's
    order (little-endian multi-word, = RADULS) permuted by hysortk_amd.paradis_order must reproduce it element for element."""
    import hysortk_amd as H
    from oracle import hsk_oracle as O
    from tests import util
    seqs = util.read_fasta(util.GOLDEN + "/reads_small.fa")
    pk, off, ln = O.pack_reads(seqs)
    r = O.count(pk, off, ln, k=51, m=17, L=1, U=65535, ntasks=5, sorter=2)
    gold = util.load_count("count_k51p.txt")
    raduls = util.result_strings(r.keys, 51)
    assert raduls != [g[0] for g in gold]                       # the two sorters do order K=51 differently
    perm = H.paradis_order(r.keys, r.cnt, r.task_off)
    assert [raduls[i] for i in perm] == [g[0] for g in gold]
    assert r.cnt[perm].tolist() == [g[1] for g in gold]


def test_every_tuning_name_the_library_reads_is_documented():
    """hsk_config::tuning names (tune("...") in hysortk_amd/csrc) against INTEGRATION.md section 5: a switch nobody can find is a switch nobody can use."""
    import glob
    import re
    names = set()
    for f in glob.glob(os.path.join(util.ROOT, "hysortk_amd", "csrc", "*")):
        names |= set(re.findall(r'tune\("([a-z0-9_]+)"', open(f).read()))
    doc = open(os.path.join(util.ROOT, "INTEGRATION.md")).read()
    assert len(names) > 30
    assert sorted(n for n in names if "`%s`" % n not in doc and "`%s=" % n not in doc) == []


def test_scripts_compile():
    """bench.py, __graft_entry__.py and everything under tools/ at least parse (they run on the GPU box only)."""
    files = [os.path.join(util.ROOT, "bench.py"), os.path.join(util.ROOT, "__graft_entry__.py")]
    for d, _, fs in os.walk(os.path.join(util.ROOT, "tools")):
        files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
    for f in files:
        compile(open(f).read(), f, "exec")
