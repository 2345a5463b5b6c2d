"""Experiment: how much do two independent pipelines gain from running concurrently on one GPU?
(upper estimate of what a two-lane batch pipeline inside one job could gain)"""
import sys, time, threading
sys.path.insert(0, ".")
import hysortk_amd as H

NREADS = int(sys.argv[1]) if len(sys.argv) > 1 else 33_333_333     # x150 = 5 Gbp per job
def make(seed):
    c = H.Context(K=31, M=17, L=15, U=40, keep_device=True)
    dp, nb, do, dl = c.synth_reads(NREADS * 150 // 30, 150, NREADS, seed)          # 30x coverage
    return c, (dp, nb, do, dl, NREADS)

jobs = [make(1), make(2)]
def run(j):
    c, a = jobs[j]
    r = c.count_device(*a)
    return r.info["total_kmers"]

for j in (0, 1): run(j)         # warm
t0 = time.time(); n = run(0) + run(1); t_seq = time.time() - t0
th = [threading.Thread(target=run, args=(j,)) for j in (0, 1)]
t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; t_con = time.time() - t0
print("k-mers %d  sequential %.1f ms  concurrent %.1f ms  gain %.2fx" % (n, t_seq * 1e3, t_con * 1e3, t_seq / t_con))
