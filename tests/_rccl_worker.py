"""Worker of tests/test_gpu_rccl.py: one of WORLD_SIZE processes.  Runs the PRODUCT path end to end -- hsk_comm_init (RCCL
communicator, or the stand-in transport named by HSK_RCCL_LIB when the ranks share one GPU; the id travels over a gloo
group) and hsk_count on this rank's share of the reads -- and dumps the rank's list.  With spec["fail"] = "<rank>:<site>"
the first count runs with that failure injected (HSK_TEST_FAIL) and must fail on EVERY rank; the count is then repeated on
the same contexts and communicator without it.  Nothing here touches the oracle: the parent test compares."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import hysortk_amd as H  # noqa: E402
from hysortk_amd import dist as hdist  # noqa: E402


def main():
    spec = json.loads(sys.argv[1])
    comm = hdist.Comm(backend="gloo")
    rank, size = comm.rank, comm.size
    seqs = json.load(open(spec["reads"]))
    counts = H.plan_partition_reads([len(s) for s in seqs], size)          # FastaIndex::getpartition
    first = int(counts[:rank].sum())
    mine = seqs[first:first + int(counts[rank])]
    dna = H.DnaBuffer.from_sequences(mine)
    rid_base = comm.exscan_sum(dna.size())
    assert rid_base == first
    device = int(os.environ.get("HSK_FORCE_DEVICE", comm.local_rank))      # several ranks on one GPU (stand-in transport)
    ctx = H.Context(K=spec["K"], M=spec["M"], L=spec["L"], U=spec["U"], EXT=spec["EXT"], ntasks=spec["ntasks"], device=device)
    ctx.comm_init(comm)
    failure = None
    if spec.get("fail"):
        os.environ["HSK_TEST_FAIL"] = spec["fail"]
        t0 = time.time()
        try:
            ctx.count(dna, rid_base=rid_base)
            failure = dict(code=0, msg="", seconds=time.time() - t0)
        except H.HskError as e:
            failure = dict(code=int(e.status), msg=str(e), seconds=time.time() - t0)
        del os.environ["HSK_TEST_FAIL"]
        comm.barrier()
    res = ctx.count(dna, rid_base=rid_base)                                # (after a failure: same context, same communicator)
    st = ctx.stats()
    out = dict(kmers=res.kmers, cnt=res.cnt, task_off=res.task_off, histo=res.histo, heavy=np.array([st["heavy_tasks"]]), combine_pairs=np.array([st["combine_pairs"]]),
               dropped=np.array([st["dropped_kmers"]]), total_kmers=np.array([res.info["total_kmers"]]))
    if failure is not None:
        out.update(fail_code=np.array([failure["code"]]), fail_seconds=np.array([failure["seconds"]]), fail_msg=np.array([failure["msg"]]))
    if spec["EXT"]:
        out.update(payload_off=res.payload_off, pos=res.pos, rid=res.rid)
    np.savez(spec["out"] % rank, **out)
    ctx.close()
    comm.barrier()
    comm.destroy()


if __name__ == "__main__":
    main()
