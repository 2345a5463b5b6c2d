import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hysortk_amd as H
rng = np.random.default_rng(91)
g = "".join(rng.choice(list("ACGT"), 20000))
reads = [g[p:p + 150] for p in rng.integers(0, len(g) - 150, 3000)]
for pre, nvar in (("AACCGGTTACGTACGGTCAA", 3200), ("ACTGACTGGTCAGTCAACGT", 16000)):
    for v in range(nvar):
        reads.append(pre + "".join(rng.choice(list("ACGT"), 40)))
    reads += reads[-50:]
dna = H.DnaBuffer.from_sequences(reads)
print("reads", len(reads), flush=True)
with H.Context(K=31, M=17, L=1, U=65535, ntasks=8) as c:
    res = c.count(dna)
    print("ok", len(res), c.stats(), flush=True)
