"""Host-side mirror of the reference's public interface (include/hysortk.hpp:10-16) on top of the
C ABI of libhsk.so.  Same names, argument meaning and error behaviour as the reference:

    read_dna_buffer(fasta_fname, comm)        reference src/hysortk.cpp:18-33
    kmer_count(mydna, comm)                   reference src/hysortk.cpp:36-96
    print_kmer_histogram(kmerlist, comm)      reference src/hysortk.cpp:98-136
    write_output_file(kmerlist, outdir, comm) reference src/hysortk.cpp:138-164
    DnaBuffer / DnaSeq                        reference include/dnabuffer.hpp:14, include/dnaseq.hpp:33
    KmerList (= KmerListS)                    reference include/kmer.hpp:368-410

The reference's compile-time macros (KMER_SIZE, MINIMIZER_SIZE, LOWER/UPPER_KMER_FREQ, EXTENSION,
Makefile:1-46) are keyword arguments with the Makefile's defaults.  `comm` is a torch.distributed
process group wrapper (hysortk_amd.dist.Comm) or None for one process; every function is
collective over it, like the reference over its MPI_Comm.  All computation happens in libhsk.so
on the GPU; this module only moves buffers and formats text.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib

# reference include/dnaseq.hpp:138-157: A/a/N/n -> 0, C/c -> 1, G/g -> 2, T/t -> 3, other -> 4
_CODETAB = np.full(256, 4, dtype=np.uint8)
for _ch, _c in (("A", 0), ("a", 0), ("N", 0), ("n", 0), ("C", 1), ("c", 1), ("G", 2), ("g", 2), ("T", 3), ("t", 3)):
    _CODETAB[ord(_ch)] = _c


class HskError(RuntimeError):
    """Raised where the reference throws std::runtime_error / aborts."""

    def __init__(self, status, detail=""):
        self.status = status
        msg = _lib.load().hsk_strerror(status).decode()
        super().__init__(msg + (": " + detail if detail else ""))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_PINNED = {}


def pinned_empty(n, dtype=np.uint8):
    """numpy array of n elements in pinned host memory (hsk_host_alloc): a DnaBuffer held in such arrays is read by the GPU
    in place (hsk_count's zero-copy ingest).  Release with pinned_free()."""
    dt = np.dtype(dtype)
    nbytes = max(int(n) * dt.itemsize, 1)
    p = _lib.load().hsk_host_alloc(nbytes)
    if not p:
        raise MemoryError("hsk_host_alloc(%d) failed" % nbytes)
    arr = np.frombuffer((C.c_uint8 * nbytes).from_address(p), dtype=dt, count=int(n))
    _PINNED[arr.ctypes.data] = p
    return arr


def pinned_free(arr):
    p = _PINNED.pop(arr.ctypes.data, None)
    if p:
        _lib.load().hsk_host_free(p)


def pack_sequence(seq):
    """2-bit packs one read exactly like DnaSeq::compress (reference src/dnaseq.cpp:9-31): base i in
    byte i/4 at shift 6-2*(i%4); tail bits zero; a non-ACGTN byte yields code 4 whose shifted value
    (truncated to 8 bits) is OR-ed in, as in the reference."""
    raw = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    n = raw.size
    nb = (n + 3) // 4
    codes = np.zeros(nb * 4, dtype=np.uint16)
    codes[:n] = _CODETAB[raw]
    codes = codes.reshape(nb, 4)
    sh = np.array([6, 4, 2, 0], dtype=np.uint16)
    return (np.bitwise_or.reduce((codes << sh) & 0xFF, axis=1)).astype(np.uint8)


class DnaSeq:
    """Non-owning view of one packed read (reference include/dnaseq.hpp:33)."""

    def __init__(self, length, mem):
        self._len, self._mem = int(length), mem

    def size(self):
        return self._len

    def numbytes(self):
        return (self._len + 3) // 4

    def data(self):
        return self._mem

    def __getitem__(self, i):
        return int(self._mem[i // 4] >> (6 - 2 * (i % 4))) & 3

    def ascii(self):
        b = np.asarray(self._mem[: self.numbytes()], dtype=np.uint8)
        codes = np.stack([(b >> 6) & 3, (b >> 4) & 3, (b >> 2) & 3, b & 3], axis=1).reshape(-1)[: self._len]
        return np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes().decode()


class DnaBuffer:
    """Owning contiguous 2-bit buffer + read index (reference include/dnabuffer.hpp:14).
    Every read starts on a byte boundary (src/dnabuffer.cpp:24-31)."""

    def __init__(self, bufsize=0):
        self._chunks = []
        self._lens = []
        self._buf = None
        self._off = None
        self._bufsize = bufsize

    @classmethod
    def from_arrays(cls, packed, read_off, read_len):
        b = cls()
        b._buf = np.ascontiguousarray(packed, dtype=np.uint8)
        b._off = np.ascontiguousarray(read_off, dtype=np.uint64)
        b._lens = list(np.asarray(read_len, dtype=np.uint32))
        return b

    @classmethod
    def from_sequences(cls, seqs):
        b = cls()
        for s in seqs:
            b.push_back(s)
        return b

    def push_back(self, s, length=None):
        if length is not None:
            s = s[:length]
        self._chunks.append(pack_sequence(s))
        self._lens.append(len(s))
        self._buf = None

    def _finalize(self):
        if self._buf is None:
            self._buf = np.concatenate(self._chunks) if self._chunks else np.zeros(0, dtype=np.uint8)
            nb = np.array([c.size for c in self._chunks], dtype=np.uint64)
            self._off = np.zeros(len(self._chunks), dtype=np.uint64)
            if len(self._chunks):
                self._off[1:] = np.cumsum(nb)[:-1]
        return self._buf

    def size(self):
        return len(self._lens)

    __len__ = size

    def getbufsize(self):
        return int(self._finalize().size)

    def __getitem__(self, i):
        buf = self._finalize()
        o = int(self._off[i])
        return DnaSeq(self._lens[i], buf[o:o + (int(self._lens[i]) + 3) // 4])

    def arrays(self):
        buf = self._finalize()
        return buf, self._off, np.asarray(self._lens, dtype=np.uint32)


class KmerList:
    """KmerListS (reference include/kmer.hpp:410): parallel arrays instead of a vector of structs.
    kmers[i] are the TKmer words (longs[0..nw)), cnt[i] the count; with EXTENSION entry i owns
    pos/rid[payload_off[i]:payload_off[i]+cnt[i]] (KmerListEntryS::pos / ::rid); see payload(i)."""

    def __init__(self, k, kmers, cnt, task_off, payload_off=None, pos=None, rid=None, histo=None, info=None):
        self.k = k
        self.kmers, self.cnt, self.task_off = kmers, cnt, task_off
        self.payload_off, self.pos, self.rid = payload_off, pos, rid
        self.histo = histo
        self.info = info or {}

    def __len__(self):
        return int(self.cnt.size)

    def payload(self, i):
        """(pos, rid) arrays of entry i (EXTENSION)."""
        a = int(self.payload_off[i]); b = a + int(self.cnt[i])
        return self.pos[a:b], self.rid[a:b]

    def kmer_string(self, i):
        w = self.kmers[i]
        return "".join("ACGT"[(int(w[j // 32]) >> (2 * (31 - j % 32))) & 3] for j in range(self.k))

    def strings(self):
        n = len(self)
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        cols = [lut[((self.kmers[:, j // 32] >> np.uint64(2 * (31 - j % 32))) & np.uint64(3)).astype(np.int64)] for j in range(self.k)]
        arr = np.stack(cols, axis=1) if n else np.zeros((0, self.k), dtype=np.uint8)
        return [r.tobytes().decode() for r in arr]


class DeviceResult:
    """A k-mer list that stays in HBM (HSK_FLAG_KEEP_DEVICE).  task(t) gives the device addresses of task t's entries and,
    with EXTENSION, its CSR payload (payload_off / pos / rid); fetch(t) copies them to numpy arrays."""

    def __init__(self, ctx, res):
        self.ctx, self.res = ctx, res
        self.ntasks, self.nw, self.n = int(res.ntasks), int(res.nw), int(res.n)
        self.task_off = np.ctypeslib.as_array(res.task_off, shape=(self.ntasks + 1,)).copy()

    def task(self, t):
        e, po, pos, rid = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, npay, base = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.ctx._check(self.ctx.lib.hsk_result_device_task(C.byref(self.res), t, C.byref(e), C.byref(n), C.byref(po), C.byref(pos), C.byref(rid), C.byref(npay), C.byref(base)))
        return dict(entries=e.value, n=int(n.value), payload_off=po.value, pos=pos.value, rid=rid.value, npay=int(npay.value), payload_base=int(base.value))

    def fetch(self, t):
        d = self.task(t)
        out = dict(n=d["n"], npay=d["npay"], payload_base=d["payload_base"])
        ent = self.ctx.d2h(d["entries"], d["n"] * (self.nw + 1) * 8).view(np.uint64).reshape(d["n"], self.nw + 1) if d["n"] else np.zeros((0, self.nw + 1), np.uint64)
        out["kmers"], out["cnt"] = ent[:, :self.nw].copy(), ent[:, self.nw].copy()
        if d["payload_off"] and d["n"]:
            out["payload_off"] = self.ctx.d2h(d["payload_off"], d["n"] * 8).view(np.uint64)
        if d["pos"] and d["npay"]:
            out["pos"] = self.ctx.d2h(d["pos"], d["npay"] * 4).view(np.uint32)
            out["rid"] = self.ctx.d2h(d["rid"], d["npay"] * 4).view(np.int32)
        return out

    def close(self):
        if self.res is not None:
            self.ctx.lib.hsk_result_free(self.ctx.h, C.byref(self.res))
            self.res = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Context:
    """One hsk_ctx (one process, one GPU)."""

    def __init__(self, K=31, M=17, L=15, U=40, EXT=0, ntasks=0, device=0, plain_dispatcher=0, radix_bits=8, profile=False, keep_device=False,
                 plan=None, tuning=None):
        """plan: None (two prefix passes + LDS aggregation, the library's choice), "no_aggregation" (HSK_FLAG_NO_AGGREGATION) or
        "full_sort" (HSK_FLAG_FULL_SORT: the reference's algorithm, LSD over all key bytes + merge-count over the sorted array)."""
        self.lib = _lib.load()
        cfg = _lib.Config()
        self.lib.hsk_config_default(C.byref(cfg))
        cfg.kmer_size, cfg.minimizer_size, cfg.lower_freq, cfg.upper_freq = K, M, L, U
        cfg.extension, cfg.ntasks, cfg.device, cfg.plain_dispatcher, cfg.radix_bits = EXT, ntasks, device, plain_dispatcher, radix_bits
        cfg.flags = (_lib.FLAG_PROFILE if profile else 0) | (_lib.FLAG_KEEP_DEVICE if keep_device else 0) | \
            {None: 0, "no_aggregation": _lib.FLAG_NO_AGGREGATION, "full_sort": _lib.FLAG_FULL_SORT, "no_combine": _lib.FLAG_NO_COMBINE}[plan]
        # tuning: dict or "name=value,..." (hsk_config::tuning): forced paths for tests and a few measured thresholds, per context
        if isinstance(tuning, dict):
            tuning = ",".join("%s=%d" % (k, int(v)) for k, v in tuning.items())
        cfg.tuning = tuning.encode() if tuning else None
        self.keep_device = bool(keep_device)
        self.cfg = cfg
        self.K, self.EXT = K, EXT
        self.nw = (K + 31) // 32
        h = C.c_void_p()
        rc = self.lib.hsk_init(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise HskError(rc)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.hsk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise HskError(rc, self.lib.hsk_last_error(self.h).decode())

    # ---- multi-GPU -----------------------------------------------------------------------------
    def comm_init(self, comm):
        """comm: hysortk_amd.dist.Comm; rank 0 creates the RCCL id, it is broadcast over comm."""
        if comm is None or comm.size == 1:
            return
        ident = (C.c_char * _lib.UNIQUE_ID_BYTES)()
        if comm.rank == 0:
            self._check(self.lib.hsk_comm_get_unique_id(ident))
        raw = comm.bcast_bytes(bytes(ident), root=0)
        buf = (C.c_char * _lib.UNIQUE_ID_BYTES).from_buffer_copy(raw)
        self._check(self.lib.hsk_comm_init(self.h, comm.size, comm.rank, buf))

    def copy_peak(self, nbytes=1 << 32, iters=3):
        """GB/s (read + written) of a plain 16-byte-per-lane HBM copy on this GPU: the measured yardstick beside the 8 TB/s spec."""
        g = C.c_double(0)
        self._check(self.lib.hsk_copy_peak(self.h, nbytes, iters, C.byref(g)))
        return float(g.value)

    def comm_selftest(self):
        """One-rank RCCL communicator on this GPU: all-reduce and grouped send/recv to self, results checked."""
        self._check(self.lib.hsk_comm_selftest(self.h))

    # ---- the hot path -----------------------------------------------------------------------------
    def _wrap(self, res):
        n, nw, nt = int(res.n), int(res.nw), int(res.ntasks)
        info = dict(total_kmers=int(res.total_kmers), total_supermers=int(res.total_supermers),
                    total_supermer_bytes=int(res.total_supermer_bytes), ntasks=nt,
                    ms_total=res.ms_total, ms_parse=res.ms_parse, ms_exchange=res.ms_exchange, ms_extract=res.ms_extract,
                    ms_sort=res.ms_sort, ms_count=res.ms_count, ms_d2h=res.ms_d2h)
        task_off = np.ctypeslib.as_array(res.task_off, shape=(nt + 1,)).copy()
        histo = np.ctypeslib.as_array(res.histo, shape=(int(res.histo_len),)).copy()
        kmers = cnt = payoff = pos = rid = None
        if res.entries:
            e = np.ctypeslib.as_array(res.entries, shape=(max(n, 1) * (nw + 1),))[: n * (nw + 1)].reshape(n, nw + 1)
            kmers, cnt = e[:, :nw].copy(), e[:, nw].copy()
            if res.payload_off:
                payoff = np.ctypeslib.as_array(res.payload_off, shape=(n + 1,)).copy()
                P = int(payoff[n])
                pos = np.ctypeslib.as_array(res.pos, shape=(max(P, 1),))[:P].copy()
                rid = np.ctypeslib.as_array(res.rid, shape=(max(P, 1),))[:P].copy()
        else:
            info["n"] = n
        self.lib.hsk_result_free(self.h, C.byref(res))
        if kmers is None:
            kmers, cnt = np.zeros((0, nw), dtype=np.uint64), np.zeros(0, dtype=np.uint64)
        return KmerList(self.K, kmers, cnt, task_off, payoff, pos, rid, histo, info)

    def count_resident(self, dna, rid_base=0):
        """kmer_count with the result LEFT IN HBM (the context must have keep_device=True): returns a DeviceResult whose
        per-task entry / CSR payload arrays a following GPU stage reads in place (hsk_result_device_task)."""
        if not self.keep_device:
            raise ValueError("count_resident needs Context(keep_device=True)")
        packed, off, lens = dna.arrays() if isinstance(dna, DnaBuffer) else dna
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        res = _lib.Result()
        self._check(self.lib.hsk_count(self.h, _p(packed), packed.size, _p(off), _p(lens), lens.size, rid_base, C.byref(res)))
        return DeviceResult(self, res)

    def count(self, dna, rid_base=0):
        packed, off, lens = dna.arrays() if isinstance(dna, DnaBuffer) else dna
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        res = _lib.Result()
        self._check(self.lib.hsk_count(self.h, _p(packed), packed.size, _p(off), _p(lens), lens.size, rid_base, C.byref(res)))
        return self._wrap(res)

    def count_host_timed(self, packed, off, lens, rid_base=0):
        """Wall time of ONE hsk_count() call on host arrays (bench.py's e2e_host leg): returns (seconds, entries, info); the
        result block is handed back to the library right away (no numpy copy: the C ABI result is the product)."""
        import time
        res = _lib.Result()
        t0 = time.perf_counter()
        rc = self.lib.hsk_count(self.h, _p(packed), packed.size, _p(off), _p(lens), lens.size, rid_base, C.byref(res))
        dt = time.perf_counter() - t0
        self._check(rc)
        n = int(res.n)
        info = dict(ms_total=res.ms_total, ms_parse=res.ms_parse, ms_extract=res.ms_extract, ms_sort=res.ms_sort, ms_count=res.ms_count, ms_d2h=res.ms_d2h,
                    total_kmers=int(res.total_kmers), ntasks=int(res.ntasks))
        self.lib.hsk_result_free(self.h, C.byref(res))
        return dt, n, info

    def count_device(self, d_packed, packed_bytes, d_off, d_len, nreads, rid_base=0):
        res = _lib.Result()
        self._check(self.lib.hsk_count_device(self.h, d_packed, packed_bytes, d_off, d_len, nreads, rid_base, C.byref(res)))
        return self._wrap(res)

    def count_loopback(self, dnas):
        """`len(dnas)` virtual ranks on this GPU (hsk_count_loopback): returns ([KmerList per rank], owner table)."""
        R = len(dnas)
        arrs = [d.arrays() if isinstance(d, DnaBuffer) else d for d in dnas]
        pk = [np.ascontiguousarray(a[0], dtype=np.uint8) for a in arrs]
        of = [np.ascontiguousarray(a[1], dtype=np.uint64) for a in arrs]
        ln = [np.ascontiguousarray(a[2], dtype=np.uint32) for a in arrs]
        PP = (C.c_void_p * R)(*[x.ctypes.data for x in pk])
        OP = (C.c_void_p * R)(*[x.ctypes.data for x in of])
        LP = (C.c_void_p * R)(*[x.ctypes.data for x in ln])
        nb = np.array([x.size for x in pk], dtype=np.uint64)
        nr = np.array([x.size for x in ln], dtype=np.uint64)
        outs = (_lib.Result * R)()
        owner = np.zeros(1024, dtype=np.int32)
        self._check(self.lib.hsk_count_loopback(self.h, R, PP, _p(nb), OP, LP, _p(nr), outs, _p(owner), owner.size))
        nt = int(outs[0].ntasks)
        res = [self._wrap(outs[r]) for r in range(R)]
        return res, owner[:nt].copy()

    def count_loopback_device(self, reads, wrap=True):
        """The same with every virtual rank's reads resident in HBM: reads = [(d_packed, packed_bytes, d_off, d_len, nreads)] as synth_reads
        returns them.  Returns ([KmerList (or DeviceResult with keep_device) per rank], owner table)."""
        R = len(reads)
        PP = (C.c_void_p * R)(*[r[0] for r in reads]); OP = (C.c_void_p * R)(*[r[2] for r in reads]); LP = (C.c_void_p * R)(*[r[3] for r in reads])
        nb = np.array([r[1] for r in reads], dtype=np.uint64); nr = np.array([r[4] for r in reads], dtype=np.uint64)
        outs = (_lib.Result * R)()
        owner = np.zeros(1024, dtype=np.int32)
        self._check(self.lib.hsk_count_loopback_device(self.h, R, PP, _p(nb), OP, LP, _p(nr), outs, _p(owner), owner.size))
        nt = int(outs[0].ntasks)
        return [self._wrap(outs[r]) for r in range(R)], owner[:nt].copy()

    def stats(self, reset=True):
        s = _lib.Stats()
        self._check(self.lib.hsk_get_stats(self.h, C.byref(s), 1 if reset else 0))
        return {k: getattr(s, k) for k, _ in _lib.Stats._fields_ if k != "reserved"}

    # ---- stages -------------------------------------------------------------------------------------
    def stage_destinations(self, dna):
        packed, off, lens = dna.arrays() if isinstance(dna, DnaBuffer) else dna
        packed = np.ascontiguousarray(packed, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.uint64); lens = np.ascontiguousarray(lens, dtype=np.uint32)
        total = int(np.maximum(lens.astype(np.int64) - self.K + 1, 0).sum())
        dest = np.zeros(max(total, 1), dtype=np.int32)
        doff = np.zeros(lens.size + 1, dtype=np.uint64)
        self._check(self.lib.hsk_stage_destinations(self.h, _p(packed), packed.size, _p(off), _p(lens), lens.size, _p(dest), total, _p(doff)))
        return dest[:total], doff

    def stage_task_kmers(self, dna, task, rid_base=0):
        packed, off, lens = dna.arrays() if isinstance(dna, DnaBuffer) else dna
        packed = np.ascontiguousarray(packed, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.uint64); lens = np.ascontiguousarray(lens, dtype=np.uint32)
        cap = int(np.maximum(lens.astype(np.int64) - self.K + 1, 0).sum())
        keys = np.zeros((max(cap, 1), self.nw), dtype=np.uint64)
        pos = np.zeros(max(cap, 1), dtype=np.uint32)
        rid = np.zeros(max(cap, 1), dtype=np.int32)
        n = C.c_uint64(0)
        self._check(self.lib.hsk_stage_task_kmers(self.h, _p(packed), packed.size, _p(off), _p(lens), lens.size, rid_base, task,
                                                  _p(keys), _p(pos), _p(rid), cap, C.byref(n)))
        n = int(n.value)
        return keys[:n], pos[:n], rid[:n]

    def stage_sort(self, keys, vals=None):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        if keys.ndim == 1:
            keys = keys.reshape(-1, 1)
        n, nw = keys.shape
        out = keys.copy()
        v = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint64).copy()
        self._check(self.lib.hsk_stage_sort(self.h, _p(out), _p(v), n, nw))
        return (out, v) if vals is not None else out

    def stage_count_sorted(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        if keys.ndim == 1:
            keys = keys.reshape(-1, 1)
        n, nw = keys.shape
        out = np.zeros((max(n, 1), nw + 1), dtype=np.uint64)
        m = C.c_uint64(0)
        self._check(self.lib.hsk_stage_count_sorted(self.h, _p(keys), n, nw, _p(out), n, C.byref(m)))
        m = int(m.value)
        return out[:m, :nw].copy(), out[:m, nw].copy()

    # ---- synthetic reads in HBM ------------------------------------------------------------------------
    def synth_reads(self, genome_len, read_len, nreads, seed, first_read=0, error_rate=0.0):
        dp, do, dl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nb = C.c_uint64(0)
        if error_rate:
            self._check(self.lib.hsk_synth_reads_err(self.h, genome_len, read_len, nreads, seed, first_read, float(error_rate), C.byref(dp), C.byref(nb), C.byref(do), C.byref(dl)))
        else:
            self._check(self.lib.hsk_synth_reads(self.h, genome_len, read_len, nreads, seed, first_read, C.byref(dp), C.byref(nb), C.byref(do), C.byref(dl)))
        return dp, int(nb.value), do, dl

    def synth_free(self, dp, do, dl):
        self.lib.hsk_synth_free(self.h, dp, do, dl)

    def format_entries(self, kmers, cnt):
        """The "KMER\\tcount\\n" lines of the entries, formatted on the GPU (hsk_format_entries); returns bytes."""
        n = len(cnt)
        if n == 0:
            return b""
        nw = kmers.shape[1]
        e = np.ascontiguousarray(np.concatenate([kmers.astype(np.uint64), np.asarray(cnt, dtype=np.uint64)[:, None]], axis=1))
        need = C.c_uint64()
        self._check(self.lib.hsk_format_entries(self.h, _p(e), n, nw, 0, None, 0, C.byref(need)))
        buf = np.zeros(int(need.value), dtype=np.uint8)
        self._check(self.lib.hsk_format_entries(self.h, _p(e), n, nw, 0, _p(buf), buf.size, C.byref(need)))
        return buf.tobytes()

    def d2h_into(self, arr, dptr, nbytes):
        self._check(self.lib.hsk_memcpy_d2h(self.h, _p(arr), dptr, nbytes))

    def d2h(self, dptr, nbytes):
        out = np.zeros(nbytes, dtype=np.uint8)
        self._check(self.lib.hsk_memcpy_d2h(self.h, _p(out), dptr, nbytes))
        return out


# ---- pure-host planning (no GPU needed) ------------------------------------------------------------------
def plan_tot_tasks(omp_max_threads, nprocs, thread_per_worker=4, avg_task_per_worker=3):
    return _lib.load().hsk_plan_tot_tasks(omp_max_threads, thread_per_worker, avg_task_per_worker, nprocs)


def plan_classify(task_kmers, unbalanced_ratio=2.3):
    a = np.ascontiguousarray(task_kmers, dtype=np.uint64)
    out = np.zeros(a.size, dtype=np.int32)
    rc = _lib.load().hsk_plan_classify(_p(a), a.size, unbalanced_ratio, _p(out))
    if rc:
        raise HskError(rc)
    return out


def plan_dispatch(task_bytes, nprocs, plain=False, upper_coe=1.5, step=0.05):
    a = np.ascontiguousarray(task_bytes, dtype=np.uint64)
    out = np.zeros(a.size, dtype=np.int32)
    rc = _lib.load().hsk_plan_dispatch(_p(a), a.size, nprocs, 1 if plain else 0, upper_coe, step, _p(out))
    if rc:
        raise HskError(rc)          # "Cannot dispatch tasks. May be too unbalanced." (kmerops.cpp:1319)
    return out


def plan_partition_reads(read_len, nprocs):
    a = np.ascontiguousarray(read_len, dtype=np.uint64)
    out = np.zeros(nprocs, dtype=np.uint64)
    rc = _lib.load().hsk_plan_partition_reads(_p(a), a.size, nprocs, _p(out))
    if rc:
        raise HskError(rc)
    return out


def plan_exchange(nranks, rank, owner, size_matrix):
    """All-to-all-v plan of the supermer exchange for `rank` (see hsk_plan_exchange in include/hsk.h).
    Returns (send_recv [nranks, 8], segs [ntasks, nranks, 4])."""
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    M = np.ascontiguousarray(size_matrix, dtype=np.uint64)
    ntasks = owner.size
    assert M.shape == (nranks, ntasks, 3)
    sr = np.zeros((nranks, 8), dtype=np.uint64)
    segs = np.zeros((ntasks, nranks, 4), dtype=np.uint64)
    rc = _lib.load().hsk_plan_exchange(nranks, rank, ntasks, _p(owner), _p(M), _p(sr), _p(segs))
    if rc:
        raise HskError(rc)
    return sr, segs


# ---- the four reference functions ---------------------------------------------------------------------------
def read_fai(path):
    """Parses <fasta>.fai records: name, length, byte offset of first base, bases per line
    (reference src/fastaindex.cpp:118-123 reads the first four columns)."""
    recs = []
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if len(p) >= 4:
                recs.append((p[0], int(p[1]), int(p[2]), int(p[3])))
    return recs


def read_fai5(path):
    """Same with the line width in bytes (column 5; the reference assumes linebases + 1, src/fastaindex.cpp:285)."""
    recs = []
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if len(p) >= 4:
                recs.append((p[0], int(p[1]), int(p[2]), int(p[3]), int(p[4]) if len(p) >= 5 else int(p[3]) + 1))
    return recs


class DeviceDna:
    """A DnaBuffer that lives in HBM (packed reads + read index), as hsk_pack_fasta / hsk_synth_reads produce it."""

    def __init__(self, ctx, dp, nbytes, do, dl, nreads, first_read_id=0):
        self.ctx, self.dp, self.nbytes, self.do, self.dl, self.nreads, self.first_read_id = ctx, dp, nbytes, do, dl, nreads, first_read_id

    def count(self, rid_base=None):
        return self.ctx.count_device(self.dp, self.nbytes, self.do, self.dl, self.nreads, rid_base=self.first_read_id if rid_base is None else rid_base)

    def count_resident_device(self, rid_base=None):
        """hsk_count_device with the result LEFT IN HBM (Context(keep_device=True)): a DeviceResult."""
        if not self.ctx.keep_device:
            raise ValueError("count_resident_device needs Context(keep_device=True)")
        res = _lib.Result()
        self.ctx._check(self.ctx.lib.hsk_count_device(self.ctx.h, self.dp, self.nbytes, self.do, self.dl, self.nreads,
                                                      self.first_read_id if rid_base is None else rid_base, C.byref(res)))
        return DeviceResult(self.ctx, res)

    def packed(self):
        return self.ctx.d2h(self.dp, self.nbytes) if self.nbytes else np.zeros(0, np.uint8)

    def free(self):
        if self.dp is not None:
            self.ctx.synth_free(self.dp, self.do, self.dl)
            self.dp = None


def read_dna_buffer_device(ctx, fasta_fname, comm=None):
    """FASTA + .fai -> this rank's reads, 2-bit packed ON THE GPU (hsk_pack_fasta): the host only maps the file.
    Same partition of the records over the ranks as read_dna_buffer."""
    recs = read_fai5(fasta_fname + ".fai")
    size = 1 if comm is None else comm.size
    rank = 0 if comm is None else comm.rank
    counts = plan_partition_reads([r[1] for r in recs], size) if size > 1 else np.array([len(recs)], dtype=np.uint64)
    first = int(counts[:rank].sum())
    mine = recs[first:first + int(counts[rank])]
    n = len(mine)
    pos = np.array([r[2] for r in mine], dtype=np.uint64)
    rlen = np.array([r[1] for r in mine], dtype=np.uint32)
    lb = np.array([r[3] for r in mine], dtype=np.uint32)
    lw = np.array([r[4] for r in mine], dtype=np.uint32)
    text = np.zeros(0, dtype=np.uint8)
    if n:
        lo = int(pos.min())
        ends = [int(p) + (((l + b - 1) // b - 1) * w + (l - ((l + b - 1) // b - 1) * b) if b and l else l) for p, l, b, w in zip(pos, rlen.astype(np.int64), lb.astype(np.int64), lw.astype(np.int64))]
        hi = max(ends)
        text = np.memmap(fasta_fname, dtype=np.uint8, mode="r")[lo:hi]
        pos = pos - np.uint64(lo)
    text = np.ascontiguousarray(text)
    dp, do, dl, nb = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
    ctx._check(ctx.lib.hsk_pack_fasta(ctx.h, _p(text) if text.size else None, text.size, _p(pos) if n else None, _p(rlen) if n else None,
                                      _p(lb) if n else None, _p(lw) if n else None, n, C.byref(dp), C.byref(nb), C.byref(do), C.byref(dl)))
    return DeviceDna(ctx, dp.value, int(nb.value), do.value, dl.value, n, first)


def read_dna_buffer(fasta_fname, comm=None):
    """FASTA + .fai -> this rank's DnaBuffer: rank r gets a contiguous run of records chosen by
    FastaIndex::getpartition (reference src/fastaindex.cpp:52-100), 2-bit packed."""
    recs = read_fai(fasta_fname + ".fai")
    size = 1 if comm is None else comm.size
    rank = 0 if comm is None else comm.rank
    counts = plan_partition_reads([r[1] for r in recs], size) if size > 1 else np.array([len(recs)], dtype=np.uint64)
    first = int(counts[:rank].sum())
    mine = recs[first:first + int(counts[rank])]
    buf = DnaBuffer()
    buf.first_read_id = first
    if not mine:
        return buf
    with open(fasta_fname, "rb") as f:
        for _, length, pos, linebases in mine:
            nlines = (length + linebases - 1) // linebases if linebases else 0
            f.seek(pos)
            raw = f.read(length + nlines)
            buf.push_back(raw.replace(b"\n", b"")[:length].decode())
    return buf


_CTX_CACHE = {}


def kmer_count(mydna, comm=None, K=31, M=17, L=15, U=40, EXT=0, ntasks=0, device=None):
    """The path the reference times as "Overall kmer counting (Excluding I/O)" (src/hysortk.cpp:91)."""
    if device is None:
        device = 0 if comm is None else comm.local_rank
    key = (K, M, L, U, EXT, ntasks, device, None if comm is None else id(comm))
    ctx = _CTX_CACHE.get(key)
    if ctx is None:
        ctx = Context(K=K, M=M, L=L, U=U, EXT=EXT, ntasks=ntasks, device=device)
        ctx.comm_init(comm)
        _CTX_CACHE[key] = ctx
    rid_base = 0
    if comm is not None and comm.size > 1:
        rid_base = comm.exscan_sum(mydna.size())        # MPI_Exscan, reference src/kmerops.cpp:65-71
    return ctx.count(mydna, rid_base=rid_base)


def histogram_text(histo):
    """Text of print_kmer_histogram (reference src/hysortk.cpp:122-131); 64-bit bins (the reference's
    int bins overflow beyond 2^31-1 k-mers per count)."""
    lines = ["#count\tnumkmers"]
    for i in range(1, len(histo)):
        if histo[i] > 0:
            lines.append("%d\t%d" % (i, int(histo[i])))
    return "\n".join(lines) + "\n\n"


def paradis_order(kmers, cnt, task_off):
    """The order a PARADIS build of the reference (SORT=1, or SORT=0 without SLURM_TASKS_PER_NODE: src/kmerops.cpp:1330-1360) leaves an
    UNFILTERED task in, for multi-word keys (K > 32; for K <= 32 both sorters give ascending u64).  paradis::sort
    (dependency/Paradis/paradissort.hpp:42-216) partitions by key byte from byte NBYTES-1 down -- the key as a little-endian multi-word
    integer, which is the order this library returns -- and hands every bucket of 2..64 k-mer INSTANCES to std::sort with operator<
    (longs[0] first, include/kmer.hpp:217), buckets of more than 64 to the next byte, nothing below byte 0.  Returns the permutation that
    turns this library's list into that order.  The bucket sizes count instances, so the order of a FILTERED list depends on k-mers the
    filter dropped and cannot be rebuilt from the list (L = 1, U = 65535 only)."""
    kmers = np.asarray(kmers, dtype=np.uint64)
    cnt = np.asarray(cnt, dtype=np.uint64)
    n, nw = kmers.shape
    perm = np.arange(n, dtype=np.int64)
    kb = np.ascontiguousarray(kmers).view(np.uint8).reshape(n, nw * 8)        # byte b of the key = column b (little endian)
    csum = np.concatenate([[0], np.cumsum(cnt.astype(np.int64))])

    def rec(lo, hi, byte):
        col = kb[lo:hi, byte]
        cuts = np.flatnonzero(col[1:] != col[:-1]) + 1
        bounds = [lo] + (cuts + lo).tolist() + [hi]
        if byte == 0:
            return
        for a, b in zip(bounds[:-1], bounds[1:]):
            c = int(csum[b] - csum[a])
            if c > 64:
                rec(a, b, byte - 1)
            elif c > 1 and b - a > 1:
                order = np.lexsort(tuple(kmers[a:b, j] for j in range(nw - 1, -1, -1)))      # operator<: longs[0] most significant
                perm[a:b] = a + order

    for t in range(len(task_off) - 1):
        a, b = int(task_off[t]), int(task_off[t + 1])
        if b - a > 1:
            rec(a, b, nw * 8 - 1)
    return perm


def print_kmer_histogram(kmerlist, comm=None, file=None):
    histo = np.asarray(kmerlist.histo, dtype=np.uint64)
    if comm is not None and comm.size > 1:
        histo = comm.allreduce_sum(histo)                # MPI_Allreduce, reference src/hysortk.cpp:115
    if comm is None or comm.rank == 0:
        (file or sys.stdout).write(histogram_text(histo))
    if comm is not None:
        comm.barrier()


def write_output_file(kmerlist, output_dir, comm=None, ctx=None):
    """<output_dir>/<rank>.out with lines "KMER\\tcount" (reference src/hysortk.cpp:138-164).  With a Context the text is
    formatted on the GPU (hsk_format_entries): spelling 10^8 k-mers on the host takes longer than counting them."""
    rank = 0 if comm is None else comm.rank
    fname = os.path.join(output_dir, "%d.out" % rank)
    try:
        f = open(fname, "wb")
    except OSError:
        raise HskError(1, "cannot open output file " + fname)
    with f:
        if ctx is not None:
            f.write(ctx.format_entries(kmerlist.kmers, kmerlist.cnt))
        else:
            strs = kmerlist.strings()
            f.write("".join("%s\t%d\n" % (s, int(c)) for s, c in zip(strs, kmerlist.cnt)).encode())
