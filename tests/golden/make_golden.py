#!/usr/bin/env python3
"""Generates tests/golden/* by RUNNING THE REAL REFERENCE (built by oracle/build_ref.sh from
/root/reference).  Runs only in the dev container; the fixtures it writes (inputs + expected
outputs, no reference source) are committed and travel to the GPU box.

  python tests/golden/make_golden.py            # (re)generate everything

Fixtures:
  murmur.json                  MurmurHash3 known answers (8/16/24-byte keys)
  stages_<variant>.json        per-read packed bytes, canonical k-mers, m-mer hashes,
                               destinations + reference supermers for tot_tasks in {5, 47}
  reads_small.fa(.fai)         adversarial + sampled reads (FASTA input of the count fixtures)
  count_<variant>.txt          raw KmerListS of the reference, 1 rank x 8 threads (tot_tasks=5)
  hist_<variant>.txt           print_kmer_histogram text of the same run
  count_k31_np<N>.txt          sorted union of the per-rank outputs, N MPI ranks
  dispatch_k31_np<N>.json      LOG=2 task sizes + task->rank table of the reference's dispatcher
"""
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
ENV = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu:/opt/conda/lib", OMP_NUM_THREADS="8")

VARIANTS = {  # name: (K, M, L, U, EXT, SORT, LOG)
    "k31": (31, 17, 1, 65535, 0, 2, 1),
    "k31ext": (31, 17, 1, 65535, 1, 2, 1),
    "k51": (51, 17, 1, 65535, 0, 2, 1),
    "k51p": (51, 17, 1, 65535, 0, 1, 1),
    "k31f": (31, 17, 3, 40, 0, 2, 1),
    "k21": (21, 9, 1, 65535, 0, 1, 1),
    "k31log": (31, 17, 1, 65535, 0, 2, 2),
    # multi-word minimizers (Mmer<2>, Mmer<3>: 16- and 24-byte murmur), added in round 2 with `--only k51m35,k77m65`
    "k51m35": (51, 35, 1, 65535, 0, 2, 1),
    "k77m65": (77, 65, 1, 65535, 0, 2, 1),
}
ROUND1 = ("k31", "k31ext", "k51", "k51p", "k31f", "k21", "k31log")


def build(v):
    K, M, L, U, EXT, SORT, LOG = VARIANTS[v]
    subprocess.check_call([os.path.join(ROOT, "oracle", "build_ref.sh"), v] + [str(x) for x in (K, M, L, U, EXT, SORT, LOG)])


def rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgtNn", "TGCAtgcaNn"))


def write_fasta(path, seqs, width=80):
    with open(path, "w") as f, open(path + ".fai", "w") as fai:
        off = 0
        for i, s in enumerate(seqs):
            hdr = ">r%d\n" % i
            f.write(hdr)
            off += len(hdr)
            fai.write("r%d\t%d\t%d\t%d\t%d\n" % (i, len(s), off, width, width + 1))
            for j in range(0, len(s), width):
                line = s[j:j + width] + "\n"
                f.write(line)
                off += len(line)


def stage_reads(rng):
    g = "".join(rng.choice(list("ACGT"), 1200))
    return [
        "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGGCTAAGCTTNACGTAACCGGTTGGCCAATTACGTAC",   # survey KAT read (70 bases)
        g,
        rc(g),
        "ACGTACGTAC",                               # shorter than every K
        "ACGTTGCAAGGCTTAACCGGTTACGATCGAT",          # exactly 31
        "acgtnnacgtNNACGTTGCAAGGCTtaaccggttacgatcgatcgggctaagcttgacgtaaccggttggccaattacgtacaacc",
        "A" * 600,                                  # one minimizer, triggers the 250-base cap
        "AC" * 200,
        "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGGCTAAGCTTGACGTAACC",  # 52 bases: 2 51-mers
        g[:300] + "T" * 90 + g[300:500],
    ]


def small_reads(rng):
    """adversarial reads + 150-bp error-free reads sampled from a 6 kbp genome (20x)."""
    G = 6000
    g = "".join(rng.choice(list("ACGT"), G))
    reads = [g[:3000], rc(g[1000:2500]), "ACGTACGTAC", g[100:131], "acgtn" * 30 + g[200:300].lower(), "A" * 200]
    n = (20 * G) // 150
    for _ in range(n):
        p = int(rng.integers(0, G - 150))
        s = g[p:p + 150]
        reads.append(rc(s) if rng.random() < 0.5 else s)
    # a few reads with odd lengths so byte-boundary padding varies
    for ln in (31, 32, 33, 51, 52, 97, 149, 151):
        p = int(rng.integers(0, G - ln))
        reads.append(g[p:p + ln])
    return reads


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw).stdout


def hist_of(stdout):
    i = stdout.index("#count\tnumkmers")
    body = stdout[i:]
    j = body.index("\n\n")
    return body[: j + 2]


def only_variants(names):
    """Fixtures of additional macro sets on the SAME inputs (stage reads re-derived from the seed, reads_small.fa as committed)."""
    rng = np.random.default_rng(20251003)
    seqs = stage_reads(rng)
    tmp = os.path.join(REF, "stage_reads.txt")
    open(tmp, "w").write("\n".join(seqs) + "\n")
    fa = os.path.join(HERE, "reads_small.fa")
    for v in names:
        build(v)
        out = run([os.path.join(REF, v, "ref_harness"), "stages", tmp, "5", "47"])
        json.dump(json.loads(out), open(os.path.join(HERE, "stages_%s.json" % v), "w"), separators=(",", ":"))
        outp = os.path.join(HERE, "count_%s.txt" % v)
        so = run([os.path.join(REF, v, "ref_harness"), "count", fa, outp])
        open(os.path.join(HERE, "hist_%s.txt" % v), "w").write(hist_of(so))
        print("wrote fixtures of", v)


def main():
    if '--only' in sys.argv:
        return only_variants(sys.argv[sys.argv.index('--only') + 1].split(","))
    only_mr = '--multirank-only' in sys.argv
    for v in ROUND1:
        build(v)
    rng = np.random.default_rng(20251003)
    fa = os.path.join(HERE, "reads_small.fa")
    if not only_mr:
        single_rank(rng, fa)
    multi_rank(fa)
    print("golden fixtures written to", HERE)


def single_rank(rng, fa):
    # ---- murmur
    out = run([os.path.join(REF, "k31", "ref_harness"), "murmur"])
    json.dump(json.loads(out), open(os.path.join(HERE, "murmur.json"), "w"), indent=0)
    # ---- stages
    seqs = stage_reads(rng)
    tmp = os.path.join(REF, "stage_reads.txt")
    open(tmp, "w").write("\n".join(seqs) + "\n")
    for v in ("k31", "k31ext", "k51", "k21"):
        out = run([os.path.join(REF, v, "ref_harness"), "stages", tmp, "5", "47"])
        json.dump(json.loads(out), open(os.path.join(HERE, "stages_%s.json" % v), "w"), separators=(",", ":"))
    # ---- full counts, 1 rank x 8 threads => tot_tasks = 8/4*3-1 = 5
    reads = small_reads(rng)
    write_fasta(fa, reads)
    for v in ("k31", "k31ext", "k51", "k51p", "k31f", "k21"):
        outp = os.path.join(HERE, "count_%s.txt" % v)
        so = run([os.path.join(REF, v, "ref_harness"), "count", fa, outp])
        open(os.path.join(HERE, "hist_%s.txt" % v), "w").write(hist_of(so))


def multi_rank(fa):
    # ---- multi-rank runs (sorted union + LOG=2 dispatch tables)
    # NB: the reference silently LOSES entries when ranks x threads oversubscribes the cores
    # (observed here: -n 3 / -n 4 with 4 threads each on 8 cores drop 50-600 of 6004 entries, run to
    # run different; 2 threads per rank never did).  The fixtures therefore use 2 threads per rank
    # (tot_tasks = 3 * nprocs, kmerops.cpp:40-43) and every run is checked against the 1-rank count.
    n_expected = len(open(os.path.join(HERE, "count_k31.txt")).read().splitlines())
    for np_ in (2, 3):
        env = dict(ENV, OMP_NUM_THREADS="2")
        outdir = os.path.join(REF, "out_np%d" % np_)
        os.makedirs(outdir, exist_ok=True)
        for attempt in range(5):
            for f in os.listdir(outdir):
                os.remove(os.path.join(outdir, f))
            p = subprocess.run(["/opt/conda/bin/mpiexec", "-n", str(np_), os.path.join(REF, "k31log", "hysortk_ref"), fa, outdir],
                               check=True, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            so = p.stdout
            lines = []
            per_rank = []
            for r in range(np_):
                rl = open(os.path.join(outdir, "%d.out" % r)).read().splitlines()
                per_rank.append(rl)
                lines += rl
            if len(lines) == n_expected:
                break
            print("reference lost entries at np=%d (%d of %d), retrying" % (np_, len(lines), n_expected))
        else:
            raise RuntimeError("reference never produced a complete result at np=%d" % np_)
        lines.sort()
        open(os.path.join(HERE, "count_k31_np%d.txt" % np_), "w").write("\n".join(lines) + "\n")
        sizes = [int(x) for x in re.findall(r"Task \d+ size: (\d+)", so)]
        table = {}
        for mm in re.finditer(r"Process (\d+): ([\d ]*)", so):
            table[int(mm.group(1))] = [int(x) for x in mm.group(2).split()]
        types = [int(x) for x in re.findall(r"Task \d+: (\d)\b", so)]
        json.dump({"nprocs": np_, "omp_threads": 2, "task_bytes": sizes, "task_ids_per_rank": table, "task_types": types,
                   "entries_per_rank": [len(x) for x in per_rank], "histogram": hist_of(so)},
                  open(os.path.join(HERE, "dispatch_k31_np%d.json" % np_), "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
