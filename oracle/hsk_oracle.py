"""ctypes binding of oracle/hsk_oracle.c (the CPU restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Parity status: pinned against the real
reference through tests/golden/ (tests/test_oracle_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhsk_oracle.so")
_SRC = os.path.join(_HERE, "hsk_oracle.c")


def build(force=False):
    """gcc-compile the C restatement (seconds)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", _SO, _SRC])
    return _SO


class _Result(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("nw", C.c_int32), ("ext", C.c_int32),
        ("keys", C.POINTER(C.c_uint64)), ("cnt", C.POINTER(C.c_uint64)),
        ("payoff", C.POINTER(C.c_uint64)), ("pos", C.POINTER(C.c_uint32)), ("rid", C.POINTER(C.c_int32)),
        ("task_off", C.POINTER(C.c_uint64)),
        ("total_kmers", C.c_uint64), ("total_supermers", C.c_uint64), ("total_supermer_bytes", C.c_uint64),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.hsko_pack.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.hsko_murmur64.restype = C.c_uint64
        L.hsko_murmur64.argtypes = [C.c_void_p, C.c_uint32]
        L.hsko_rep_mers.restype = C.c_int64
        L.hsko_rep_mers.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.hsko_mmer_hashes.restype = C.c_int64
        L.hsko_mmer_hashes.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.hsko_dests.restype = C.c_int64
        L.hsko_dests.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.hsko_supermers.restype = C.c_int64
        L.hsko_supermers.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.hsko_copy_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.hsko_cnt_bytes.restype = C.c_int
        L.hsko_cnt_bytes.argtypes = [C.c_int]
        L.hsko_classify.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        L.hsko_dispatch_balanced.restype = C.c_int
        L.hsko_dispatch_balanced.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.hsko_dispatch_roundrobin.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.hsko_tot_tasks.restype = C.c_int
        L.hsko_tot_tasks.argtypes = [C.c_int] * 4
        L.hsko_count.restype = C.c_int
        L.hsko_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(_Result)]
        L.hsko_result_free.argtypes = [C.POINTER(_Result)]
        L.hsko_histogram_text.restype = C.c_size_t
        L.hsko_histogram_text.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.hsko_mer_string.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(seq):
    """ASCII read -> 2-bit packed bytes (DnaSeq::compress)."""
    if isinstance(seq, str):
        seq = seq.encode()
    out = np.zeros((len(seq) + 3) // 4, dtype=np.uint8)
    lib().hsko_pack(seq, len(seq), _p(out))
    return out


def pack_reads(seqs):
    """list of ASCII reads -> (packed uint8[], read_off uint64[], read_len uint32[]); every read
    starts on a byte boundary (DnaBuffer::push_back, src/dnabuffer.cpp:24)."""
    parts = [pack(s) for s in seqs]
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    nb = np.array([len(p) for p in parts], dtype=np.uint64)
    off = np.zeros(len(seqs), dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum(nb)[:-1]
    packed = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(packed), off, lens


def murmur64(words):
    a = np.ascontiguousarray(np.asarray(words, dtype=np.uint64))
    return int(lib().hsko_murmur64(_p(a), 8 * a.size))


def rep_mers(packed, length, k):
    nw = (k + 31) // 32
    n = max(0, length - k + 1)
    out = np.zeros((n, nw), dtype=np.uint64)
    if n:
        p = np.ascontiguousarray(packed)
        lib().hsko_rep_mers(_p(p), length, k, _p(out))
    return out


def mmer_hashes(packed, length, m):
    n = max(0, length - m + 1)
    out = np.zeros(n, dtype=np.uint64)
    if n:
        p = np.ascontiguousarray(packed)
        lib().hsko_mmer_hashes(_p(p), length, m, _p(out))
    return out


def dests(packed, length, k, m, tot_tasks):
    n = max(0, length - k + 1)
    out = np.zeros(n, dtype=np.int32)
    if n:
        p = np.ascontiguousarray(packed)
        lib().hsko_dests(_p(p), length, k, m, tot_tasks, _p(out))
    return out


def supermers(dest, k, packed=None):
    """Reference supermer split of one read: list of (task, start, len[, bytes])."""
    dest = np.ascontiguousarray(dest, dtype=np.int32)
    n = dest.size
    t = np.zeros(max(n, 1), dtype=np.int32)
    s = np.zeros(max(n, 1), dtype=np.uint32)
    l = np.zeros(max(n, 1), dtype=np.uint32)
    ns = lib().hsko_supermers(_p(dest), n, k, _p(t), _p(s), _p(l))
    res = []
    for i in range(ns):
        if packed is not None:
            nb = lib().hsko_cnt_bytes(int(l[i]))
            b = np.zeros(nb, dtype=np.uint8)
            pk = np.ascontiguousarray(packed)
            lib().hsko_copy_bits(_p(b), _p(pk), int(s[i]), int(l[i]))
            res.append((int(t[i]), int(s[i]), int(l[i]), b))
        else:
            res.append((int(t[i]), int(s[i]), int(l[i])))
    return res


def classify(task_kmers, ratio=2.3):
    a = np.ascontiguousarray(task_kmers, dtype=np.uint64)
    out = np.zeros(a.size, dtype=np.int32)
    lib().hsko_classify(_p(a), a.size, ratio, _p(out))
    return out


def dispatch_balanced(task_bytes, nprocs, upper=1.5, step=0.05):
    a = np.ascontiguousarray(task_bytes, dtype=np.uint64)
    out = np.zeros(a.size, dtype=np.int32)
    rc = lib().hsko_dispatch_balanced(_p(a), a.size, nprocs, upper, step, _p(out))
    if rc != 0:
        raise RuntimeError("Cannot dispatch tasks. May be too unbalanced.")
    return out


def tot_tasks(omp_threads, nprocs, thread_per_worker=4, avg_task_per_worker=3):
    return lib().hsko_tot_tasks(omp_threads, thread_per_worker, avg_task_per_worker, nprocs)


class CountResult:
    def __init__(self, keys, cnt, task_off, payoff=None, pos=None, rid=None, stats=None):
        self.keys, self.cnt, self.task_off = keys, cnt, task_off
        self.payoff, self.pos, self.rid = payoff, pos, rid
        self.stats = stats or {}


def count(packed, read_off, read_len, k=31, m=17, L=1, U=65535, ext=0, ntasks=5, rid_base=0,
          task_owner=None, my_rank=0, sorter=2, fast=False):
    """Whole path (supermer split -> extract -> sort -> count -> filter), raw reference order."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    read_len = np.ascontiguousarray(read_len, dtype=np.uint32)
    res = _Result()
    owner = None
    if task_owner is not None:
        owner = np.ascontiguousarray(task_owner, dtype=np.int32)
    rc = lib().hsko_count(_p(packed), _p(read_off), _p(read_len), read_len.size, k, m, L, U, ext, ntasks, rid_base,
                          _p(owner) if owner is not None else None, my_rank, sorter, 1 if fast else 0, C.byref(res))
    if rc != 0:
        raise ValueError("hsko_count: bad arguments")
    n, nw = res.n, res.nw
    keys = np.ctypeslib.as_array(res.keys, shape=(max(n, 1) * nw,))[: n * nw].reshape(n, nw).copy()
    cnt = np.ctypeslib.as_array(res.cnt, shape=(max(n, 1),))[:n].copy()
    task_off = np.ctypeslib.as_array(res.task_off, shape=(ntasks + 1,)).copy()
    payoff = pos = rid = None
    if ext:
        payoff = np.ctypeslib.as_array(res.payoff, shape=(n + 1,)).copy()
        P = int(payoff[n])
        pos = np.ctypeslib.as_array(res.pos, shape=(max(P, 1),))[:P].copy()
        rid = np.ctypeslib.as_array(res.rid, shape=(max(P, 1),))[:P].copy()
    stats = dict(total_kmers=res.total_kmers, total_supermers=res.total_supermers,
                 total_supermer_bytes=res.total_supermer_bytes)
    lib().hsko_result_free(C.byref(res))
    return CountResult(keys, cnt, task_off, payoff, pos, rid, stats)


def task_digests(packed, read_off, read_len, k=31, m=17, ext=0, ntasks=5, rid_base=0, task_sel=None):
    """(n[ntasks], mix[ntasks]): number of k-mer instances per task and the sum of digest_mix over them (hsko_task_digests):
    streaming, for inputs far too large for count()."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    read_off = np.ascontiguousarray(read_off, dtype=np.uint64)
    read_len = np.ascontiguousarray(read_len, dtype=np.uint32)
    n = np.zeros(ntasks, dtype=np.uint64); mix = np.zeros(ntasks, dtype=np.uint64)
    sel = None if task_sel is None else np.ascontiguousarray(task_sel, dtype=np.uint8)
    lib().hsko_task_digests.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib().hsko_task_digests(_p(packed), _p(read_off), _p(read_len), read_len.size, k, m, ext, ntasks, rid_base, _p(sel) if sel is not None else None, _p(n), _p(mix))
    if rc != 0:
        raise ValueError("hsko_task_digests: bad arguments")
    return n, mix


def _fmix64(x):
    x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xff51afd7ed558ccd); x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xc4ceb9fe1a85ec53)
    return x ^ (x >> np.uint64(33))


def digest_mix(keys, pos=None, rid=None):
    """numpy twin of hsko_digest_mix: keys uint64 [n, nw] (+ per-row pos / rid for EXTENSION) -> uint64 [n]."""
    keys = np.asarray(keys, dtype=np.uint64)
    x = np.zeros(keys.shape[0], dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(keys.shape[1]):
            x = _fmix64(x ^ (keys[:, j] + np.uint64((0x9e3779b97f4a7c15 * (j + 1)) & 0xFFFFFFFFFFFFFFFF)))
        if pos is not None:
            pr = (np.asarray(rid).astype(np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF)) << np.uint64(32) | np.asarray(pos, dtype=np.uint64)
            x = _fmix64(x ^ _fmix64(pr + np.uint64(0x632be59bd9b4e019)))
    return x


def entries_digests(keys, cnt, task_off, payload=None):
    """Per-task (n, mix) of a result list: n = sum of cnt, mix = sum of cnt * digest_mix(key) (mod 2^64); payload = (payload_off, pos,
    rid) with EXTENSION: mix = sum over every payload of digest_mix(key, pos, rid)."""
    nt = len(task_off) - 1
    n = np.zeros(nt, dtype=np.uint64); mix = np.zeros(nt, dtype=np.uint64)
    cnt = np.asarray(cnt, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for t in range(nt):
            a, b = int(task_off[t]), int(task_off[t + 1])
            if a == b:
                continue
            n[t] = cnt[a:b].sum(dtype=np.uint64)
            if payload is None:
                mix[t] = (digest_mix(keys[a:b]) * cnt[a:b]).sum(dtype=np.uint64)
            else:
                po, pos, rid = payload
                p0, p1 = int(po[a]), int(po[b])
                rep = np.repeat(np.asarray(keys[a:b], dtype=np.uint64), cnt[a:b].astype(np.int64), axis=0)
                mix[t] = digest_mix(rep, pos[p0:p1], rid[p0:p1]).sum(dtype=np.uint64)
    return n, mix


def histogram_text(cnt):
    a = np.ascontiguousarray(cnt, dtype=np.uint64)
    need = lib().hsko_histogram_text(_p(a), a.size, None, 0)
    buf = C.create_string_buffer(need + 1)
    lib().hsko_histogram_text(_p(a), a.size, buf, need + 1)
    return buf.value.decode()


def mer_string(words, k):
    a = np.ascontiguousarray(words, dtype=np.uint64)
    buf = C.create_string_buffer(k + 1)
    lib().hsko_mer_string(_p(a), k, buf)
    return buf.value.decode()


def string_to_words(s, k=None):
    """ACGT string -> left-aligned 2-bit words (Kmer::set_kmer(char const*), kmer.hpp:188-207)."""
    k = k or len(s)
    nw = (k + 31) // 32
    w = [0] * nw
    for i, ch in enumerate(s[:k]):
        w[i // 32] |= "ACGT".index(ch) << (2 * (31 - i % 32))
    return np.array(w, dtype=np.uint64)
