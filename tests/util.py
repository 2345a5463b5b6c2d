"""Shared helpers for the tests (fixture parsing, canonical forms)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

VARIANTS = {  # name: dict(K, M, L, U, ext, sorter)   -- the macro sets the golden files were generated with
    "k31": dict(k=31, m=17, L=1, U=65535, ext=0, sorter=2),
    "k31ext": dict(k=31, m=17, L=1, U=65535, ext=1, sorter=2),
    "k51": dict(k=51, m=17, L=1, U=65535, ext=0, sorter=2),
    "k51p": dict(k=51, m=17, L=1, U=65535, ext=0, sorter=1),
    "k31f": dict(k=31, m=17, L=3, U=40, ext=0, sorter=2),
    "k21": dict(k=21, m=9, L=1, U=65535, ext=0, sorter=1),
    "k51m35": dict(k=51, m=35, L=1, U=65535, ext=0, sorter=2),      # two-word minimizers (Mmer<2>, 16-byte murmur)
    "k77m65": dict(k=77, m=65, L=1, U=65535, ext=0, sorter=2),      # three-word k-mers and minimizers (24-byte murmur)
}


def read_fasta(path):
    """Minimal FASTA reader: list of sequences (line breaks removed)."""
    seqs, cur = [], None
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if cur is not None:
                    seqs.append("".join(cur))
                cur = []
            elif cur is not None:
                cur.append(line)
    if cur is not None:
        seqs.append("".join(cur))
    return seqs


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_count(name):
    """Golden raw KmerListS: list of (kmer string, cnt, pos list | None, rid list | None) in file order."""
    out = []
    with open(os.path.join(GOLDEN, name)) as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) == 2:
                out.append((parts[0], int(parts[1]), None, None))
            else:
                pos = [int(x) for x in parts[2].split(",")] if parts[2] else []
                rid = [int(x) for x in parts[3].split(",")] if parts[3] else []
                out.append((parts[0], int(parts[1]), pos, rid))
    return out


def words_to_str(words, k):
    s = []
    for i in range(k):
        w = int(words[i // 32])
        s.append("ACGT"[(w >> (2 * (31 - i % 32))) & 3])
    return "".join(s)


def result_strings(keys, k):
    """uint64 [n, nw] -> list of k-mer strings."""
    keys = np.asarray(keys, dtype=np.uint64)
    n, nw = keys.shape
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    cols = []
    for i in range(k):
        w = keys[:, i // 32]
        cols.append(lut[((w >> np.uint64(2 * (31 - i % 32))) & np.uint64(3)).astype(np.int64)])
    arr = np.stack(cols, axis=1) if n else np.zeros((0, k), dtype=np.uint8)
    return [row.tobytes().decode() for row in arr]


def hex_words(lst):
    return np.array([int(x, 16) for x in lst], dtype=np.uint64)


# Round 4: the library's test switches are per-context tuning names (hsk_config::tuning / the environment's HSK_TUNING string), no longer one
# environment variable each.  Tests written as {"HSK_PARSE_FAST": "0"} keep reading that way: tune_env() folds every such key into HSK_TUNING
# (the names below are NOT tuning names: transport hooks, diagnostics, harness variables).
_NOT_TUNING = {"HSK_TEST_PLAN", "HSK_RCCL_LIB", "HSK_RCCL_MSG_MAX", "HSK_TEST_FAIL", "HSK_FORCE_DEVICE", "HSK_FAKERCCL_TIMEOUT", "HSK_LIB", "HSK_TIMING", "HSK_BACKTRACE",
               "HSK_HOST_INGEST", "HSK_TEST_FORCE_FAKERCCL", "HSK_TUNING", "HSK_GPUS_PER_NODE"}


def tune_env(env):
    out, items = {}, []
    for k, v in (env or {}).items():
        if k.startswith("HSK_") and k not in _NOT_TUNING:
            items.append("%s=%s" % (k[4:].lower(), v))
        else:
            out[k] = v
    if items:
        out["HSK_TUNING"] = ",".join(items)
    return out


def tuning(env):
    """the same as a Context(tuning=...) string"""
    return tune_env(env).get("HSK_TUNING")
