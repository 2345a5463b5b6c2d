"""Per-kernel profile driver of the several-rank data path on one GPU: R virtual ranks through hsk_count_loopback_device, nothing else.
   rocprofv3 --kernel-trace --stats -d gpurun_out/mrp -o mrp -- python3 tools/exp/mr_profile.py [K] [bp_per_rank] [tuning]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hysortk_amd as H

K = int(sys.argv[1]) if len(sys.argv) > 1 else 31
BP = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000_000
TUNE = sys.argv[3] if len(sys.argv) > 3 else None
R, RL, COV = 8, 150, 32
G = R * BP // COV
NR = BP // RL
ctx = H.Context(K=K, M=17, L=15, U=40, ntasks=320, profile=True, keep_device=True, tuning=TUNE)
reads = []
for r in range(R):
    dp, nb, do, dl = ctx.synth_reads(G, RL, NR, 20251010, first_read=r * NR)
    reads.append((dp, nb, do, dl, NR))
for it in range(2):
    ctx.stats(reset=True)
    t0 = time.perf_counter()
    res, owner = ctx.count_loopback_device(reads)
    w = time.perf_counter() - t0
    st = ctx.stats(reset=True)
    ms = [dict(r_.info) for r_ in res]
    print("call", it, "wall %.1f ms" % (w * 1e3), "combine", st["combine_launches"], {k[3:]: round(v, 2) for k, v in ms[0].items() if k.startswith("ms_")}, flush=True)
    del res
ctx.close()
