"""Writes a FASTA of `gbp` Gbp (150-base reads, 60-base lines) + .fai into /tmp, builds examples/hysortk_main.cpp and runs it with HSK_TIMING=1:
the shim's read_dna_buffer phase by phase.   python tools/exp/fasta_ingest_time.py [gbp]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
RL, LB = 150, 60
rng = np.random.default_rng(12)
genome = rng.integers(0, 4, 40_000_000, dtype=np.uint8)
nreads = int(gbp * 1e9) // RL
path = "/tmp/ingest_%g.fa" % gbp
nl = (RL + LB - 1) // LB
w = 3 + RL + nl
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
with open(path, "wb") as f, open(path + ".fai", "w") as g:
    step = 1 << 19
    for a in range(0, nreads, step):
        n = min(step, nreads - a)
        st = rng.integers(0, genome.size - RL, n)
        rec = np.empty((n, w), dtype=np.uint8)
        rec[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
        bases = lut[genome[st[:, None] + np.arange(RL)[None, :]]]
        o = 3
        for l in range(nl):
            m = min(LB, RL - l * LB)
            rec[:, o:o + m] = bases[:, l * LB:l * LB + m]; rec[:, o + m] = 10; o += m + 1
        f.write(rec.tobytes())
        g.write("".join("r\t%d\t%d\t%d\t%d\n" % (RL, i * w + 3, LB, LB + 1) for i in range(a, a + n)))
exe = "/tmp/hysortk_ingest"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-DKMER_SIZE=31", "-DMINIMIZER_SIZE=17", "-DLOWER_KMER_FREQ=2", "-DUPPER_KMER_FREQ=60",
                       "-DEXTENSION=0", "-o", exe, os.path.join(ROOT, "examples", "hysortk_main.cpp"), "-L", os.path.join(ROOT, "hysortk_amd"), "-lhsk", "-pthread",
                       "-Wl,-rpath," + os.path.join(ROOT, "hysortk_amd")])
for it in range(2):
    t0 = time.time()
    out = subprocess.run([exe, path], env=dict(os.environ, HSK_TIMING="1"), capture_output=True, text=True)
    print("run", it, "%.2f s wall" % (time.time() - t0))
    print("\n".join(l for l in out.stderr.splitlines() if "shim" in l))
    print("\n".join(l for l in out.stdout.splitlines() if "read_dna_buffer" in l))
print("file bytes", os.path.getsize(path))
