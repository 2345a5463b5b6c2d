"""The combining extraction (hysortk_amd/csrc/hsk_combine.h: supermers ordered by minimizer bucket, k-mers counted in LDS tables
where they are extracted, {k-mer, count} pairs through one scatter pass and the weighted finish) against the instance path and
the oracle.  Its switches are per-context tuning names (hsk_config::tuning, round 4), so every case runs IN THIS PROCESS with a
context of its own (rounds 2-3 forked a worker per case: the switches were environment variables read once per process);
combine_min_bytes=0 lets inputs of test size take the plan (the library's own limit is 64 MB of packed reads)."""
import numpy as np
import pytest

from tests import util
from tests import _combine_worker as W

pytestmark = pytest.mark.gpu
BASE = dict(K=31, M=17, L=2, U=200, ntasks=16, genome=1500000, read_len=150, nreads=400000, seed=77, calls=["device"])


def run(spec, env):
    """one context with the tuning `env` spells (old environment-variable names, folded by util.tuning), one result dict per call"""
    extra = spec.get("tuning_extra")
    tun = util.tuning(env)
    return W.run_spec(dict(spec, tuning=(tun + "," + extra if tun else extra) if extra else tun))


@pytest.fixture(scope="module")
def instance():
    """digest of the base input on the instance path (HSK_COMBINE=0)"""
    r = run(BASE, {"HSK_COMBINE": "0"})[0]
    assert r["combine_launches"] == 0 and r["instance_extractions"] > 0 and r["entries"] > 100000
    return r


@pytest.mark.parametrize("env,why", [
    ({}, "defaults"),
    ({"HSK_COMBINE_BUCKET": "1000000000"}, "one bucket per virtual task: the table fills and is written out again and again inside a bucket, k-mers leave in partial pairs"),
    ({"HSK_COMBINE_BUCKET": "300"}, "as many buckets as the order allows: nearly empty tables"),
    ({"HSK_COMBINE_PREFIX": "16"}, "bins of the instance path's width"),
    ({"HSK_COMBINE_PREFIX": "11"}, "few, long bins: the ladder of the weighted finish"),
    ({"HSK_SCAN_PLACE": "1"}, "items placed by scan_kernel itself into (XCD, virtual task) chunk lists: no tile records, no placement kernel"),
    ({"HSK_SCAN_PLACE": "1", "HSK_COMBINE_BUCKET": "300"}, "... with nearly empty tables"),
])
def test_combining_extraction_equals_instance_path(instance, env, why):
    r = run(BASE, dict(env, HSK_COMBINE_MIN_BYTES="0"))[0]
    assert r["combine_launches"] > 0 and r["instance_extractions"] == 0, why
    assert r["combine_kmers"] == r["total_kmers"] and 0 < r["combine_pairs"] <= r["combine_kmers"], why
    assert (r["digest"], r["entries"]) == (instance["digest"], instance["entries"]), why


def test_combining_extraction_vs_oracle(tmp_path):
    """the list itself, k-mer by k-mer, against the CPU oracle (K = 21 and 29 take the kernel's generic instance, M = 11 a wide window)"""
    from oracle import hsk_oracle as O
    for K, M, nt in ((31, 17, 8), (21, 11, 24), (29, 19, 40)):
        dump = str(tmp_path / ("c%d.npz" % K))
        spec = dict(BASE, K=K, M=M, ntasks=nt, L=1, U=65535, genome=400000, nreads=60000, dump=dump)
        r = run(spec, {"HSK_COMBINE_MIN_BYTES": "0"})[0]
        assert r["combine_launches"] > 0
        z = np.load(dump)
        want = O.count(z["packed"], z["off"], z["lens"], k=K, m=M, L=1, U=65535, ntasks=nt, fast=True)
        assert np.array_equal(want.task_off, z["task_off"]) and np.array_equal(want.keys, z["kmers"]) and np.array_equal(want.cnt, z["cnt"]), (K, M, nt)


def test_input_without_copies_leaves_the_combining_extraction():
    """uniform random reads: as many pairs as k-mers.  The first call finishes on the pairs (correct, slow) and switches the context
    to the instance path; the second call takes it.  Same list both times."""
    spec = dict(BASE, L=1, U=65535, error_rate=0.75, nreads=150000, calls=["device", "device"])
    a, b = run(spec, {"HSK_COMBINE_MIN_BYTES": "0"})
    ref = run(dict(spec, calls=["device"]), {"HSK_COMBINE": "0"})[0]
    assert a["combine_launches"] > 0 and a["combine_pairs"] * 16 > a["combine_kmers"]
    assert b["combine_launches"] == 0
    assert a["digest"] == b["digest"] == ref["digest"] and a["entries"] == ref["entries"]


@pytest.mark.parametrize("env,spec,why", [
    ({"HSK_XCD_BATCH": "0"}, dict(), "the one-task-per-XCD kernels are switched off: no batch to run the combining extraction on; the call starts again without it"),
    ({"HSK_PARSE_REC_CAP": "200"}, dict(), "tiles beyond the record capacity: the parse leaves its fast path, and the virtual tasks with it"),
    ({"HSK_SCAN_PLACE": "1", "HSK_BIN_CAP_PCT": "1"}, dict(genome=6000000, nreads=1500000), "items placed by the scan, and their chunk store runs out (a first chunk per bin and next to nothing beyond): error bit, the call again on the instance path"),
    ({"HSK_SCAN_PLACE": "1", "HSK_BIN_VMAX": "1"}, dict(genome=6000000, nreads=1500000), "items placed by the scan, a bin of more chunks than its map has entries"),
])
def test_calls_that_start_again_without_the_combining_extraction(env, spec, why):
    sp = dict(BASE, **spec)
    ref = run(sp, {"HSK_COMBINE": "0"})[0]
    r = run(sp, dict(env, HSK_COMBINE_MIN_BYTES="0"))[0]
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"]), why
    assert r["instance_extractions"] > 0, why                     # (the instance path's extraction kernel ran: the second attempt)


def test_one_large_task_overflows_its_tables_and_leaves_the_plan():
    """ONE task of 420 M k-mers has twice the k-mers per bucket the bucket order aims at (2^14 buckets per task at most): ~1200 distinct
    k-mers per bucket overflow the 2048-slot tables, k-mers leave in partial pairs (more than one pair per sixteen k-mers; bins of the finish may overflow
    on top: the call then goes round again), and the context takes the instance path from the next call on.  Same list both times."""
    sp = dict(BASE, ntasks=1, genome=20000000, nreads=3500000, L=1, U=65535, calls=["device", "device"])
    a, b = run(sp, {"HSK_COMBINE_MIN_BYTES": "0"})
    ref = run(dict(sp, calls=["device"]), {"HSK_COMBINE": "0"})[0]
    assert b["combine_launches"] == 0 and b["instance_extractions"] > 0      # (the first call ends on the pairs or goes round again: either way the plan is off afterwards)
    for r in (a, b):
        assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"])


@pytest.mark.parametrize("ntasks", [200, 500, 1000])
def test_many_tasks(ntasks):
    """200 tasks: two virtual tasks each (at most 768 in all); 500: none, the bucket order does all the bits (and the parse scan takes its
    three-kernel form from 128 columns on, with or without the combining extraction); 1000: more than the item placement takes, the instance path"""
    sp = dict(BASE, ntasks=ntasks, genome=3000000, nreads=800000)
    ref = run(sp, {"HSK_COMBINE": "0"})[0]
    r = run(sp, {"HSK_COMBINE_MIN_BYTES": "0"})[0]
    if ntasks <= 768:
        assert r["combine_launches"] > 0 and r["instance_extractions"] == 0
    else:                                                     # (more tasks than the item placement's LDS has room for: the instance path)
        assert r["combine_launches"] == 0 and r["instance_extractions"] > 0
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"])


def test_two_tasks_are_padded_to_a_batch():
    """an item-mode store pads any task count to whole batches (the instance path would take two tasks one by one)"""
    sp = dict(BASE, ntasks=2)
    ref = run(sp, {"HSK_COMBINE": "0"})[0]
    r = run(sp, {"HSK_COMBINE_MIN_BYTES": "0"})[0]
    assert r["combine_launches"] == 1 and (r["digest"], r["entries"]) == (ref["digest"], ref["entries"])


def test_host_paths_through_the_combining_extraction(instance):
    """hsk_count() from pageable and from pinned host memory (slab ingest pipelined with scan and item placement: the store is laid out
    [slab][virtual task]) give the list of the device-resident call"""
    spec = dict(BASE, genome=3000000, nreads=1100000, calls=["device", "host", "pinned", "pinned"])      # 41 MB of packed reads: above the slab-ingest limit
    rs = run(spec, {"HSK_COMBINE_MIN_BYTES": "0"}) + run(spec, {"HSK_COMBINE_MIN_BYTES": "0", "HSK_SCAN_PLACE": "1"})      # (... and with the items placed by the scan, slab by slab)
    ref = run(dict(spec, calls=["device"]), {"HSK_COMBINE": "0"})[0]
    assert all(r["combine_launches"] > 0 for r in rs)
    assert {r["digest"] for r in rs} == {ref["digest"]}


# ---- the plan is chosen inside the call (estimate_plan, hsk_api.hip): ONE call on a fresh context, library defaults (no HSK_COMBINE_MIN_BYTES) ----
BIG = dict(BASE, ntasks=0, L=1, U=65535, genome=8000000, nreads=1800000, seed=91)            # 68 MB of packed reads, 32x: the combining extraction's own limit is 64 MB


@pytest.mark.parametrize("how", ["device", "pinned", "host"])
def test_first_call_on_clean_deep_reads_takes_the_combining_extraction(how):
    r = run(dict(BIG, calls=[how]), {})[0]
    ref = run(dict(BIG, calls=["device"]), {"HSK_COMBINE": "0"})[0]
    assert r["combine_launches"] > 0 and r["instance_extractions"] == 0
    assert r["combine_pairs"] * 16 < r["combine_kmers"]
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"])


@pytest.mark.parametrize("spec,why", [
    (dict(error_rate=0.003), "0.3 % substitution errors: one k-mer in eleven is an error k-mer that occurs once"),
    (dict(error_rate=0.01), "1 % substitution errors"),
    (dict(error_rate=0.75), "uniform random reads: every k-mer once"),
    (dict(genome=64000000), "coverage 4: a k-mer has four copies"),
])
@pytest.mark.parametrize("how", ["device", "pinned"])
def test_first_call_never_starts_a_plan_the_input_does_not_pay_for(spec, why, how):
    """A real client calls kmer_count() once (reference src/hysortk.cpp:36-96): the FIRST call on a fresh context must already take the
    instance path for inputs with too few copies per k-mer -- combine_kernel is never launched -- and give the same list."""
    if how == "pinned" and spec.get("error_rate") == 0.75:
        pytest.skip("uniform reads (as many entries as k-mers: 12 s per case) once, from device memory")
    sp = dict(BIG, calls=[how], **spec)
    r = run(sp, {})[0]
    ref = run(dict(sp, calls=["device"]), {"HSK_COMBINE": "0", "HSK_PLAN_SAMPLE": "0"})[0]
    assert r["combine_launches"] == 0 and r["instance_extractions"] > 0, why
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"]), why


def test_alternating_inputs_on_one_context_choose_per_call():
    """clean, noisy, clean on ONE context: every call chooses for its own input (the context's memory of the previous call does not decide)"""
    import hysortk_amd as H
    with H.Context(K=31, M=17, L=1, U=65535, profile=True) as ctx:
        clean = ctx.synth_reads(8000000, 150, 1800000, 5)
        noisy = ctx.synth_reads(8000000, 150, 1800000, 6, error_rate=0.01)
        seen = []
        for dp, nb, do, dl in (clean, noisy, clean, noisy):
            ctx.stats(reset=True)
            r = ctx.count_device(dp, nb, do, dl, 1800000)
            st = ctx.stats(reset=True)
            seen.append((int(st["combine_launches"]) > 0, len(r)))
            del r
        for d in (clean, noisy):
            ctx.synth_free(d[0], d[2], d[3])
    assert [s[0] for s in seen] == [True, False, True, False], seen
    assert seen[0][1] == seen[2][1] and seen[1][1] == seen[3][1]


# ---- several ranks: the owner of a task builds the items from the supermers the exchange delivered (hsk_combine.h, 1b) ----------------
@pytest.mark.parametrize("R,ntasks,env,why", [
    (2, 16, {}, "two virtual ranks, one batch each"),
    (4, 64, {}, "four ranks, two task groups each: the grouped exchange feeds the batches"),
    (8, 72, {}, "eight ranks, nine tasks each: a partial batch padded"),
    (3, 24, {"HSK_COMBINE_BUCKET": "300"}, "three ranks, nearly empty tables"),
    (4, 64, {"K": 51}, "two-word keys (configs[3]'s shape) on four ranks: items of ten k-mers built by the owners, combine2_kernel"),
    (2, 2, {"HSK_COMBINE_PREFIX": "9"}, "one task per rank in 512 bins of ~6000 pairs: bins beyond the weighted finish's last table; those tasks take the weighted long way (full-width passes over the pairs + sums of equal keys), nobody starts again"),
])
def test_owner_side_combining_extraction_equals_instance_path(R, ntasks, env, why):
    sp = dict(BASE, ntasks=ntasks, genome=2000000, nreads=480000, L=1, U=65535, calls=["loopback:%d" % R])
    env = dict(env)
    if "K" in env:
        sp["K"] = env.pop("K")
    if ntasks == 2:
        sp.update(genome=8000000, nreads=800000)
    ref = run(sp, {"HSK_COMBINE": "0"})[0]
    r = run(sp, dict(env, HSK_COMBINE_MIN_BYTES="0"))[0]
    assert ref["combine_launches"] == 0 and r["combine_launches"] > 0, why
    assert r["instance_extractions"] == 0 or "HSK_COMBINE_PREFIX" in env, why        # (the weighted long way's full-width passes count their histogram launches there)
    assert r["combine_kmers"] == r["total_kmers"] == ref["total_kmers"], why
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"]), why
    if "HSK_COMBINE_PREFIX" in env:
        assert r["redone_tasks"] > 0, why


def test_weighted_long_way_on_one_gpu():
    """ONE task in 512 bins of ~2900 pairs: bins beyond the weighted finish's last table.  Rounds 2-3 started the call again on the instance path;
    now the task's pairs are sorted and summed (the call goes on): combine_kernel ran, no instance extraction, same list."""
    sp = dict(BASE, L=1, ntasks=1)
    ref = run(sp, {"HSK_COMBINE": "0"})[0]
    r = run(sp, {"HSK_COMBINE_MIN_BYTES": "0", "HSK_COMBINE_PREFIX": "9"})[0]
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"])
    assert r["combine_launches"] > 0 and r["redone_tasks"] > 0


# ---- two-word keys through the combining extraction (40 <= K <= 55): items of at most 61 - K k-mers, combine2_kernel, weighted two-word finish ----
@pytest.mark.parametrize("K,env,why", [
    (51, {}, "BASELINE configs[3]'s key shape: items of at most 10 k-mers"),
    (51, {"HSK_COMBINE_BUCKET": "1000000000"}, "one bucket per virtual task: tables dumped inside a bucket, partial pairs"),
    (51, {"HSK_COMBINE_BUCKET": "300"}, "nearly empty tables"),
    (51, {"HSK_SCAN_PLACE": "1"}, "items placed by the scan"),
    (41, {}, "9 bases in the second word: 16-k-mer items"),
    (45, {}, "items of 16 k-mers exactly (61 - 45)"),
    (55, {}, "items of six k-mers: the largest K the plan takes"),
    (47, {"HSK_SCAN_GENERIC": "1"}, "generic scan instance, items of 14"),
])
def test_two_word_keys_through_the_combining_extraction(K, env, why):
    sp = dict(BASE, K=K, L=1, U=65535)
    ref = run(sp, {"HSK_COMBINE": "0"})[0]
    r = run(sp, dict(env, HSK_COMBINE_MIN_BYTES="0"))[0]
    assert ref["combine_launches"] == 0 and r["combine_launches"] > 0 and r["instance_extractions"] == 0, why
    assert r["combine_kmers"] == r["total_kmers"] == ref["total_kmers"] and 0 < r["combine_pairs"] <= r["combine_kmers"], why
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"]), why


def test_two_word_combining_extraction_vs_oracle(tmp_path):
    from oracle import hsk_oracle as O
    for K, M, nt in ((51, 17, 8), (43, 21, 24)):
        dump = str(tmp_path / ("c%d.npz" % K))
        spec = dict(BASE, K=K, M=M, ntasks=nt, L=1, U=65535, genome=400000, nreads=60000, dump=dump)
        r = run(spec, {"HSK_COMBINE_MIN_BYTES": "0"})[0]
        assert r["combine_launches"] > 0
        z = np.load(dump)
        want = O.count(z["packed"], z["off"], z["lens"], k=K, m=M, L=1, U=65535, ntasks=nt, fast=True)
        assert np.array_equal(want.task_off, z["task_off"]) and np.array_equal(want.keys, z["kmers"]) and np.array_equal(want.cnt, z["cnt"]), (K, M, nt)


def test_pair_stores_that_run_over_are_enlarged_not_abandoned():
    """The combining extraction's buffers are sized for the pairs the call's sketch promises (four times over), not for the k-mers.  Stores that
    run over after all (here: forced to 100 000 records per task) set an error bit -- the chunk behind the store takes the overrun -- and the call
    runs once more with full-sized buffers: still the combining extraction, same list."""
    ref = run(BASE, {"HSK_COMBINE": "0"})[0]
    r = run(BASE, {"HSK_COMBINE_MIN_BYTES": "0", "HSK_PAIR_CAP_RECORDS": "100000"})[0]
    assert r["combine_launches"] > 0 and r["instance_extractions"] == 0
    assert (r["digest"], r["entries"]) == (ref["digest"], ref["entries"])


@pytest.mark.parametrize("K,U", [(31, 65535), (31, 200), (51, 65535)])
def test_a_minimizer_that_very_many_supermers_share_is_cut_into_work_units(K, U):
    """5 % of the reads are all-A reads (homopolymers, satellites: ONE k-mer and ONE minimizer bucket hold 6 % of all k-mer instances -- 20 000
    reads x 120 k-mers = 150 000 items in one bucket of the order).  The bucket is counted as slices of CB_UNIT = 8192 items by different
    workgroups whose pairs the weighted finish adds up: same list as the instance path, the all-A k-mer's count (U = 65535: capped by the
    16-bit count as in the reference; U = 200: filtered) included; the plan's sketch takes a run of one k-mer as one insert."""
    spec = dict(BASE, K=K, U=U, L=2, poly_a_pct=5.0, calls=["pinned", "host"], ntasks=8)
    a = run(spec, {"HSK_COMBINE": "0"})
    b = run(spec, {"HSK_COMBINE_MIN_BYTES": "0"})
    assert all(x["combine_launches"] == 0 for x in a) and all(x["combine_launches"] > 0 and x["instance_extractions"] == 0 for x in b)
    assert len({(x["digest"], x["entries"]) for x in a + b}) == 1 and a[0]["entries"] > 100000


@pytest.mark.parametrize("K", [31, 51])
def test_a_bin_of_very_many_records_is_counted_in_slices(K):
    """The instance path on reads of which 5 % are all-A reads: the all-A 31-mer's 2.4 M records share ONE prefix bin of one task.  The bin is cut
    into slices that many workgroups count into a table in global memory (hsk_agg.h: AggLarge), its own workgroup counts that table: same list
    as with the slices switched off (agg_large=0) and as the combining extraction."""
    spec = dict(BASE, K=K, U=65535, L=2, poly_a_pct=5.0, calls=["pinned"], ntasks=8)
    a = run(spec, {"HSK_COMBINE": "0"})[0]
    b = run(dict(spec, tuning_extra="agg_large=0"), {"HSK_COMBINE": "0"})[0]
    c_ = run(spec, {"HSK_COMBINE_MIN_BYTES": "0"})[0]
    assert a["combine_launches"] == 0 and b["combine_launches"] == 0 and c_["combine_launches"] > 0
    assert (a["digest"], a["entries"]) == (b["digest"], b["entries"]) == (c_["digest"], c_["entries"]) and a["entries"] > 100000


@pytest.mark.parametrize("K,EXT,U", [(31, 0, 40), (51, 0, 40), (31, 1, 40), (31, 0, 65535)])
def test_kmers_that_are_certain_to_be_dropped_are_left_out_by_the_scan(K, EXT, U):
    """5 % all-A reads in an input the plan's sketch looks at (68 MB of packed reads): the sample alone holds more than U (and more than 2^16) copies
    of the all-A k-mer, so it cannot be in the result and the scan clears the positions that hold it from its valid mask (scan_kernel<.., DROP>:
    no supermers, no records, no bucket or bin of millions, no task several times the others' size).  Same list -- and with EXTENSION the same
    payloads -- as with drop_certain=0; total_kmers still counts the instances; U = 65535: the sample's 110 000 copies are certain as well."""
    spec = dict(BIG, K=K, EXT=EXT, L=2, U=U, poly_a_pct=5.0, calls=["pinned"])
    a = run(spec, {})[0]
    b = run(dict(spec, tuning_extra="drop_certain=0"), {})[0]
    assert a["dropped_kmers"] == int(1800000 * 0.05) * (150 - K + 1) and b["dropped_kmers"] == 0
    assert (a["digest"], a["entries"], a["total_kmers"]) == (b["digest"], b["entries"], b["total_kmers"]) and a["entries"] > 100000


@pytest.mark.parametrize("K,EXT,R,extra,why", [
    (31, 0, 2, "", "two virtual ranks, combining extraction on the owners' side"),
    (51, 0, 2, "", "two-word keys"),
    (31, 1, 2, "", "payloads: no plan to choose, the sketch still runs for the drops"),
    (31, 0, 2, "parse_rec_cap=200", "the scan's record store runs over: the general parse kernels take over and honour the same mask"),
])
def test_certain_drops_with_several_ranks(K, EXT, R, extra, why):
    """Several ranks leave the same k-mers out or none does: the mask is the OR of the ranks' (run_pipeline: in the plan's all-reduce; the
    virtual ranks share the first one's), and a rank whose scan falls back to the general parse kernels applies it there.  Same lists as with
    drop_certain=0, total_kmers (summed over the ranks) counts the instances that were left out."""
    spec = dict(BIG, K=K, EXT=EXT, L=2, U=40, poly_a_pct=5.0, calls=["hostloop:%d" % R])
    a = run(dict(spec, tuning_extra=extra) if extra else spec, {})[0]
    b = run(dict(spec, tuning_extra="drop_certain=0"), {})[0]
    assert a["dropped_kmers"] == int(1800000 * 0.05) * (150 - K + 1) and b["dropped_kmers"] == 0, why
    assert (a["digest"], a["entries"], a["total_kmers"]) == (b["digest"], b["entries"], b["total_kmers"]) and a["entries"] > 100000, why


def test_certain_drops_through_the_general_parse_on_one_gpu():
    """parse_rec_cap=200 makes the scan's stores run over on one GPU as well (pinned reads: the pipelined ingest falls back, the call starts again
    without virtual tasks, the scan overflows again): parse_kernel<COUNT/FILL> leave the all-A positions out and count them."""
    spec = dict(BIG, K=31, L=2, U=40, poly_a_pct=5.0, calls=["pinned"])
    a = run(dict(spec, tuning_extra="parse_rec_cap=200"), {})[0]
    b = run(dict(spec, tuning_extra="drop_certain=0"), {})[0]
    c_ = run(dict(spec, tuning_extra="parse_rec_cap=200,drop_certain=0"), {})[0]
    assert a["dropped_kmers"] == int(1800000 * 0.05) * 120 and b["dropped_kmers"] == 0
    assert (a["digest"], a["entries"], a["total_kmers"]) == (b["digest"], b["entries"], b["total_kmers"])
    # (round 4 found a fault here: with most tiles beyond the capacity the supermers counted exceed the pipelined ingest's store -- its placement
    #  kernels, launched before the host sees the scan's verdict, now place nothing once a tile has overflowed)
    assert (c_["digest"], c_["entries"], c_["total_kmers"]) == (b["digest"], b["entries"], b["total_kmers"])


def test_the_certain_drops_of_a_call_do_not_outlive_it():
    """A call that left the all-A k-mer out, then the stage entry points on the same context (they parse without a sketch): every instance of the
    all-A k-mer is there again -- the call's mask and count are cleared when it returns."""
    import hysortk_amd as H
    from hysortk_amd import synth
    seqs = ["A" * 150] * 700 + list(synth.reads(200000, 150, 20000, 41))
    dna = H.DnaBuffer.from_sequences(seqs)
    with H.Context(K=31, M=17, L=2, U=40, ntasks=5, profile=True, tuning="plan_min_input=1") as c:
        r = c.count(dna)
        assert int(c.stats()["dropped_kmers"]) == 700 * 120 and int(r.info["total_kmers"]) == len(seqs) * 120
        n = sum(len(c.stage_task_kmers(dna, t)[0]) for t in range(5))
        assert n == len(seqs) * 120
        r2 = c.count(dna)
        assert int(r2.info["total_kmers"]) == len(seqs) * 120 and np.array_equal(r2.kmers, r.kmers) and np.array_equal(r2.cnt, r.cnt)
