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
