"""Worker of tests/test_multirank_cpu.py: one of WORLD_SIZE gloo processes.

Exercises the N>1 host path of the product on CPU: hysortk_amd.dist.Comm (Exscan, bcast, allreduce),
read partitioning, task dispatch and the all-to-all-v exchange plan (hsk_plan_*), with the exchange
itself carried by torch.distributed all_to_all_single over gloo in place of RCCL send/recv.  The
GPU kernels are replaced by the oracle (this is a test: only tests may use the oracle) which plays
the role of parse (supermer store per task) and of extract+sort+count on the received segments.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hysortk_amd as H  # noqa: E402
from hysortk_amd import dist as hdist  # noqa: E402
from oracle import hsk_oracle as O  # noqa: E402
from tests import util  # noqa: E402

K, M = 31, 17


def main():
    out_path = sys.argv[1]
    comm = hdist.Comm(backend="gloo")
    rank, size = comm.rank, comm.size
    import torch
    import torch.distributed as dist

    # ---- partition of the reads (FastaIndex::getpartition) + global read ids (MPI_Exscan)
    dna = H.read_dna_buffer(util.GOLDEN + "/reads_small.fa", comm)
    nreads = dna.size()
    rid_base = comm.exscan_sum(nreads)
    assert rid_base == dna.first_read_id
    packed, off, lens = dna.arrays()
    ntasks = H.plan_tot_tasks(2, size)                       # 2 threads per rank, as in the golden runs

    # ---- "parse": per-task supermer stores of this rank (oracle = stand-in for the GPU kernels)
    per_task = [[] for _ in range(ntasks)]                   # (len, bytes, pos, rid) per supermer
    for r in range(nreads):
        pk = packed[int(off[r]):]
        d = O.dests(pk, int(lens[r]), K, M, ntasks)
        for task, start, ln, by in O.supermers(d, K, pk):
            nb = (ln + 3) // 4
            per_task[task].append((ln, by[:nb].tobytes(), start, rid_base + r))
    tot = np.zeros((ntasks, 3), dtype=np.uint64)
    for t in range(ntasks):
        tot[t] = (len(per_task[t]), sum(len(x[1]) for x in per_task[t]), sum(x[0] - K + 1 for x in per_task[t]))

    # ---- dispatch on global task sizes (allreduce = MPI_Reduce + Bcast of the reference)
    task_bytes = comm.allreduce_sum(tot[:, 1] + tot[:, 0])
    owner = H.plan_dispatch(task_bytes, size)
    # ---- size matrix (every rank's row), exchange plan
    Mloc = np.zeros((size, ntasks, 3), dtype=np.uint64)
    Mloc[rank] = tot
    Mall = comm.allreduce_sum(Mloc.reshape(-1)).reshape(size, ntasks, 3)
    sr, segs = H.plan_exchange(size, rank, owner, Mall)

    # ---- storage order: tasks grouped by owner, ascending id -> send arrays
    order = sorted(range(ntasks), key=lambda t: (owner[t], t))
    slen = np.concatenate([np.array([x[0] for x in per_task[t]], dtype=np.uint8) for t in order] + [np.zeros(0, np.uint8)])
    sbytes = np.frombuffer(b"".join(b"".join(x[1] for x in per_task[t]) for t in order), dtype=np.uint8)
    assert slen.size == int(sr[:, 0].sum()) and sbytes.size == int(sr[:, 1].sum())
    for q in range(size):                                    # offsets are the storage-order prefix sums
        assert int(sr[q, 2]) == int(sr[:q, 0].sum()) and int(sr[q, 3]) == int(sr[:q, 1].sum())

    def a2a(send, scount, rcount):
        recv = torch.zeros(int(rcount.sum()), dtype=torch.uint8)
        dist.all_to_all_single(recv, torch.from_numpy(send.copy()), [int(x) for x in rcount], [int(x) for x in scount])
        return recv.numpy()

    rlen = a2a(slen, sr[:, 0], sr[:, 4])
    rbytes = a2a(sbytes, sr[:, 1], sr[:, 5])

    # ---- "extract + sort + count" of every owned task from its (task, src) segments
    lines = []
    entries = 0
    for t in range(ntasks):
        if owner[t] != rank:
            continue
        mers = []
        for p in range(size):
            so, ns, bo, ko = (int(x) for x in segs[t, p])
            b = bo
            for s in range(so, so + ns):
                ln = int(rlen[s])
                nb = (ln + 3) // 4
                mers.append(O.rep_mers(np.frombuffer(rbytes[b:b + nb].tobytes() + b"\0" * 8, dtype=np.uint8), ln, K)[:, 0])
                b += nb
        if not mers:
            continue
        allk = np.sort(np.concatenate(mers))
        u, c = np.unique(allk, return_counts=True)
        entries += u.size
        for s_, c_ in zip(util.result_strings(u.reshape(-1, 1), K), c.tolist()):
            lines.append("%s\t%d" % (s_, c_))
    hist = comm.allreduce_sum(np.bincount(np.array([int(l.split("\t")[1]) for l in lines], dtype=np.int64), minlength=400)[:400])
    json.dump({"rank": rank, "lines": lines, "entries": entries, "owner": [int(x) for x in owner], "hist": [int(x) for x in hist],
               "nreads": nreads, "rid_base": rid_base}, open(out_path, "w"))
    comm.barrier()
    comm.destroy()


if __name__ == "__main__":
    main()
