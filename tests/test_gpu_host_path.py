"""hsk_count() from host memory: pinned input read in place by the scan (zero-copy ingest), pageable input (staged copy),
device-side validation of the read index, early result copies -- all must give the list the device-resident path gives."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload(H, n=400000):
    from hysortk_amd import synth
    return synth.packed_reads(3000000, 150, n, 5)


def test_pinned_zero_copy_equals_pageable_and_device_paths(monkeypatch):
    import hysortk_amd as H
    packed, off, lens = _workload(H)                           # 15 MB: below the zero-copy threshold unless forced
    nb = (150 + 3) // 4
    reps = 3                                                   # > 16 MB of packed reads: the pinned path reads them in place
    packed = np.tile(packed, reps); lens = np.tile(lens, reps)
    off = np.arange(lens.size, dtype=np.uint64) * np.uint64(nb)
    pp, po, pl = H.pinned_empty(packed.size, np.uint8), H.pinned_empty(off.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
    pp[:] = packed; po[:] = off; pl[:] = lens
    with H.Context(K=31, M=17, L=2, U=200, ntasks=16) as c:
        a = c.count((packed, off, lens))                       # pageable: staged copies
        b = c.count((pp, po, pl))                              # pinned: scan_kernel reads the host buffer, writes the HBM copy
        st = c.stats()
        b2 = c.count((pp, po, pl))                             # again: pinned result block reused, early copies sized from the last call
    H.pinned_free(pp); H.pinned_free(po); H.pinned_free(pl)
    assert len(a) > 100000
    for x in (b, b2):
        assert np.array_equal(a.kmers, x.kmers) and np.array_equal(a.cnt, x.cnt) and np.array_equal(a.task_off, x.task_off) and np.array_equal(a.histo, x.histo)
    assert st["h2d_bytes"] > 2 * packed.size and st["d2h_bytes"] > 0


def test_invalid_read_index_is_rejected_on_the_device():
    import hysortk_amd as H
    from hysortk_amd import synth
    n = (1 << 20) + 1000                                       # large enough for the device-side check
    packed, off, lens = synth.packed_reads(500000, 150, n, 9)
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=8) as c:
        good = c.count((packed, off, lens))
        bad = off.copy(); bad[n // 2] = bad[n // 2 - 1]        # overlaps its predecessor
        with pytest.raises(H.HskError):
            c.count((packed, bad, lens))
        bad = lens.copy(); bad[-1] = 4000                      # leaves the packed buffer
        with pytest.raises(H.HskError):
            c.count((packed, off, bad))
        again = c.count((packed, off, lens))                   # the context stays usable
    assert np.array_equal(good.kmers, again.kmers) and np.array_equal(good.cnt, again.cnt)


def test_pinned_index_past_the_buffer_is_rejected():
    """The same two bad indices from PINNED arrays (offsets derived from the lengths on the device, the caller's offsets compared on
    host threads): a back-to-back index whose last read runs past packed_bytes used to pass that comparison and be counted
    truncated; it must be rejected like the pageable one."""
    import hysortk_amd as H
    from hysortk_amd import synth
    n = (1 << 20) + 1000
    packed, off, lens = synth.packed_reads(500000, 150, n, 9)             # 38 bytes per read: 40 MB, zero-copy ingest
    pp, po, pl = H.pinned_empty(packed.size, np.uint8), H.pinned_empty(off.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
    pp[:] = packed; po[:] = off; pl[:] = lens
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=8) as c:
        good = c.count((pp, po, pl))
        pl[-1] = 4000                                            # leaves the packed buffer (the offsets still lie back to back)
        with pytest.raises(H.HskError):
            c.count((pp, po, pl))
        pl[-1] = lens[-1]
        po[n // 2] = po[n // 2 - 1]                              # overlaps its predecessor
        with pytest.raises(H.HskError):
            c.count((pp, po, pl))
        po[:] = off
        again = c.count((pp, po, pl))                            # the context stays usable
        ref = c.count((packed, off, lens))
    for y in (pp, po, pl):
        H.pinned_free(y)
    assert np.array_equal(good.kmers, again.kmers) and np.array_equal(good.cnt, again.cnt)
    assert np.array_equal(good.kmers, ref.kmers) and np.array_equal(good.cnt, ref.cnt)


def test_pinned_input_with_gaps_between_reads():
    """hsk_count() on a pinned buffer derives the read offsets from the read lengths on the device (only the lengths travel ahead of
    the scan) and compares them with the caller's offsets afterwards; a buffer whose reads do NOT lie back to back (the C ABI allows
    gaps) must be noticed and counted with the caller's offsets -- same list as the gap-free buffer."""
    import hysortk_amd as H
    from hysortk_amd import synth
    n = (1 << 20) + 5000                                       # >= 2^20 reads: device-side index handling
    packed, off, lens = synth.packed_reads(2000000, 150, n, 21)            # 38 bytes per read: 40 MB
    nb = 38
    gap_at, gap = n // 3, 24
    packed_g = np.concatenate([packed[:gap_at * nb], np.full(gap, 0xFF, np.uint8), packed[gap_at * nb:]])
    off_g = off.copy(); off_g[gap_at:] += np.uint64(gap)
    arrs = []
    for pk, of in ((packed, off), (packed_g, off_g)):
        pp, po, pl = H.pinned_empty(pk.size, np.uint8), H.pinned_empty(of.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
        pp[:] = pk; po[:] = of; pl[:] = lens
        arrs.append((pp, po, pl))
    with H.Context(K=31, M=17, L=2, U=200, ntasks=16) as c:
        a = c.count(arrs[0])
        b = c.count(arrs[1])
        ref = c.count((packed, off, lens))                     # pageable: plain copies, host-side... device-side index check
    for x in arrs:
        for y in x:
            H.pinned_free(y)
    assert len(ref) > 100000
    for x in (a, b):
        assert np.array_equal(ref.kmers, x.kmers) and np.array_equal(ref.cnt, x.cnt) and np.array_equal(ref.task_off, x.task_off)


@pytest.mark.parametrize("K,EXT", [(31, 0), (51, 0), (31, 1)])
def test_slab_ingest_equals_reads_in_place(K, EXT):
    """Pinned input above 32 MB: the packed reads arrive as DMA copies slab by slab while the scan hashes the slabs before
    (h2d_slabs: 16 by default, 3 and 8 here as well; the last slab is short and a read straddles every slab edge); =0: the scan reads
    the host buffer in place as in round 2.  Same list, same histogram, and equal to the pageable path.  One context per setting, in this
    process (the switches are per-context tuning names since round 4; rounds 2 - 3 started a process per setting)."""
    import hashlib
    import hysortk_amd as H
    from hysortk_amd import synth
    from tests import util
    n = (1 << 20) + 77777
    packed, off, lens = synth.packed_reads(2000000, 150, n, 21)
    pp, po, pl = H.pinned_empty(packed.size, np.uint8), H.pinned_empty(off.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
    pp[:] = packed; po[:] = off; pl[:] = lens
    outs = []
    # ({}: ingest, scan and placement as one pipeline, the store laid out [slab][task]; ingest_pipeline=0: slab ingest, one placement;
    #  a record capacity of 300 makes some tile overflow: both fall back to the general parse kernels with the reads already in HBM)
    envs = ({}, {"HSK_H2D_SLABS": "3"}, {"HSK_H2D_SLABS": "0"}, {"HSK_H2D_SLABS": "8", "HSK_PARSE_REC_CAP": "300"}, {"HSK_INGEST_PIPELINE": "0"},
            {"HSK_INGEST_PIPELINE": "0", "HSK_PARSE_REC_CAP": "300"})
    if EXT:
        envs = envs[:1] + envs[2:3]                              # (payloads take the slab ingest without the placement pipeline: default and in-place suffice)
    try:
        for env in envs:
            with H.Context(K=K, M=17, L=2, U=200, EXT=EXT, ntasks=16, tuning=util.tuning(env)) as c:
                for src in ((pp, po, pl), (packed, off, lens), (pp, po, pl)):
                    r = c.count(src)
                    pay = b""
                    if r.pos is not None:                    # (the payloads of a k-mer come in any order: an order-independent digest per list)
                        x = r.pos.astype(np.uint64) | (r.rid.astype(np.uint64) << np.uint64(32))
                        with np.errstate(over="ignore"):
                            pay = np.array([(x * np.uint64(0x9E3779B97F4A7C15)).sum(dtype=np.uint64), np.bitwise_xor.reduce(x), x.size], dtype=np.uint64).tobytes()
                    outs.append((hashlib.sha256(r.kmers.tobytes() + r.cnt.tobytes() + r.task_off.tobytes() + r.histo.tobytes() + pay).hexdigest(), len(r)))
                    del r
    finally:
        for x in (pp, po, pl):
            H.pinned_free(x)
    assert len(outs) == 3 * len(envs) and len({o[0] for o in outs}) == 1, outs
    assert int(outs[0][1]) > 100000


def test_result_outlives_its_context():
    """hsk_destroy() before hsk_result_free(NULL, &res): the result's pinned blocks belong to the result (include/hsk.h); they must
    stay readable after the context is gone and be released exactly once."""
    import ctypes as C
    import hysortk_amd as H
    from hysortk_amd import _lib, synth
    packed, off, lens = synth.packed_reads(200000, 150, 20000, 4)
    with H.Context(K=31, M=17, L=1, U=65535, ntasks=5) as c:
        want = c.count((packed, off, lens))
    L = _lib.load()
    ctx = H.Context(K=31, M=17, L=1, U=65535, ntasks=5)
    res = _lib.Result()
    assert L.hsk_count(ctx.h, packed.ctypes.data_as(C.c_void_p), packed.size, off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), lens.size, 0, C.byref(res)) == 0
    ctx.close()                                                   # hsk_destroy
    n = int(res.n)
    e = np.ctypeslib.as_array(res.entries, shape=(n * 2,)).reshape(n, 2).copy()
    histo = np.ctypeslib.as_array(res.histo, shape=(int(res.histo_len),)).copy()
    L.hsk_result_free(None, C.byref(res))
    assert not res.entries and int(res.n) == 0
    assert np.array_equal(e[:, 0], want.kmers[:, 0]) and np.array_equal(e[:, 1], want.cnt) and np.array_equal(histo, want.histo)


def test_pinned_reads_of_almost_uniform_length():
    """Pinned input whose first, middle and last read are equally long: the device generates the read lengths from that sample instead of
    copying them, host threads verify every length while the GPU scans.  Here 1 % of the reads are shorter (none of the sampled ones):
    the guess is wrong, the call must notice and count with the real lengths -- same list as from pageable memory.  With truly uniform
    lengths (second half) the guess holds."""
    import hysortk_amd as H
    from hysortk_amd import synth
    n = (1 << 20) + 4321
    seqs_packed, off0, lens0 = synth.packed_reads(2000000, 150, n, 33)
    rng = np.random.default_rng(3)
    lens = lens0.copy()
    short = rng.choice(np.arange(1, n - 1), n // 100, replace=False)
    short = short[short != n // 2]
    lens[short] = 120                                            # the packed bytes stay where they are: a read simply ends earlier ...
    packed = seqs_packed
    off = off0                                                   # ... so the buffer has gaps too (38 bytes per slot, 30 used)
    pp, po, pl = H.pinned_empty(packed.size, np.uint8), H.pinned_empty(off.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
    pp[:] = packed; po[:] = off; pl[:] = lens
    with H.Context(K=31, M=17, L=2, U=200, ntasks=16) as c:
        a = c.count((pp, po, pl))
        ref = c.count((packed, off, lens))
        pl[:] = lens0
        b = c.count((pp, po, pl))
        ref_b = c.count((packed, off0, lens0))
        st = c.stats()
    for y in (pp, po, pl):
        H.pinned_free(y)
    assert len(ref) > 100000 and int(ref.cnt.sum()) < int(ref_b.cnt.sum())             # (the shorter reads really count for less)
    assert np.array_equal(ref.kmers, a.kmers) and np.array_equal(ref.cnt, a.cnt) and np.array_equal(ref.task_off, a.task_off)
    assert np.array_equal(ref_b.kmers, b.kmers) and np.array_equal(ref_b.cnt, b.cnt)


@pytest.mark.parametrize("ntasks", [1, 5, 7])
def test_pipelined_ingest_with_few_tasks(ntasks):
    """Pinned input above 32 MB with the reference's small task counts: the store is laid out [slab][task], every task has one segment
    per slab, and one to seven tasks take the single-task extraction / the padded batch -- same list as from pageable memory."""
    import hysortk_amd as H
    from hysortk_amd import synth
    n = (1 << 20) + 99
    packed, off, lens = synth.packed_reads(1500000, 150, n, 8)
    pp, po, pl = H.pinned_empty(packed.size, np.uint8), H.pinned_empty(off.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
    pp[:] = packed; po[:] = off; pl[:] = lens
    with H.Context(K=31, M=17, L=2, U=300, ntasks=ntasks) as c:
        a = c.count((pp, po, pl))
        ref = c.count((packed, off, lens))
    for y in (pp, po, pl):
        H.pinned_free(y)
    assert len(ref) > 100000
    assert np.array_equal(ref.kmers, a.kmers) and np.array_equal(ref.cnt, a.cnt) and np.array_equal(ref.task_off, a.task_off) and np.array_equal(ref.histo, a.histo)


def test_pinned_mixed_lengths_with_the_general_parse_k51_m35():
    """M = 35 (> SCAN_MAX_M) takes the general parse kernels from the start, which never read the host threads' verdict on a derived
    read index: hsk_count() must not derive the index there.  Pinned reads whose first, middle and last lengths agree while 1 % of the
    others are shorter (the sample says 'fixed length', which is wrong): same list as from pageable memory."""
    import hysortk_amd as H
    from hysortk_amd import synth
    n = (1 << 20) + 777
    packed, off, lens0 = synth.packed_reads(2000000, 150, n, 35)
    rng = np.random.default_rng(5)
    lens = lens0.copy()
    short = rng.choice(np.arange(1, n - 1), n // 100, replace=False)
    short = short[short != n // 2]
    lens[short] = 111
    pp, po, pl = H.pinned_empty(packed.size, np.uint8), H.pinned_empty(off.size, np.uint64), H.pinned_empty(lens.size, np.uint32)
    pp[:] = packed; po[:] = off; pl[:] = lens
    with H.Context(K=51, M=35, L=2, U=200, ntasks=16) as c:
        a = c.count((pp, po, pl))
        ref = c.count((packed, off, lens))
        pl[:] = lens0
        b = c.count((pp, po, pl))
        ref_b = c.count((packed, off, lens0))
    for y in (pp, po, pl):
        H.pinned_free(y)
    assert len(ref) > 100000 and int(ref.cnt.sum()) < int(ref_b.cnt.sum())
    assert np.array_equal(ref.kmers, a.kmers) and np.array_equal(ref.cnt, a.cnt) and np.array_equal(ref.task_off, a.task_off)
    assert np.array_equal(ref_b.kmers, b.kmers) and np.array_equal(ref_b.cnt, b.cnt)
