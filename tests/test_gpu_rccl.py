"""The real N>1 transport: two fresh processes, one GPU each, form an RCCL communicator (hsk_comm_init) and run hsk_count
on their halves of the reads -- grouped ncclSend/ncclRecv per task group on the second stream (post_exchange,
csrc/hsk_comm.h), all-reduces of task sizes / size matrix, heavy-hitter list exchange.  Per-rank lists must equal what
the virtual-rank driver (hsk_count_loopback, device copies in place of RCCL) gives on one GPU, and their union the
reference's 2-rank output.  Needs two GPUs: SKIPPED (not passed) on a one-GPU box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


def _ngpu():
    import torch
    return torch.cuda.device_count()


def _run_ranks(world, spec, tmp_path, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.ROOT, "tests", "_rccl_worker.py"), json.dumps(spec)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(spec["out"] % r) for r in range(world)]


def _split(H, seqs, R):
    counts = H.plan_partition_reads([len(s) for s in seqs], R)
    parts, first = [], 0
    for r in range(R):
        parts.append(seqs[first:first + int(counts[r])])
        first += int(counts[r])
    return parts


def _heavy_reads():
    from hysortk_amd import synth
    rng = np.random.default_rng(5)
    unit = "ACGGTCATTGCA"
    seqs = list(synth.reads(80000, 150, 6000, 31)) + [(unit * 13)[:150]] * 2500 + [(unit[5:] + unit[:5]) * 12 + "ACGTAC"] * 500
    return [seqs[i] for i in rng.permutation(len(seqs))]


CASES = {
    "golden_k31": dict(K=31, M=17, L=1, U=65535, EXT=0, ntasks=None),
    "ext": dict(K=31, M=17, L=2, U=50, EXT=1, ntasks=24),
    "groups_k51": dict(K=51, M=17, L=2, U=50, EXT=0, ntasks=40),
    "heavy_k31": dict(K=31, M=17, L=2, U=65535, EXT=0, ntasks=24),
    "heavy_k51": dict(K=51, M=17, L=2, U=65535, EXT=0, ntasks=24),
}


@pytest.mark.parametrize("case", list(CASES))
def test_two_ranks_over_rccl(case, tmp_path):
    if _ngpu() < 2:
        pytest.skip("needs two GPUs (this box shows %d): the RCCL exchange between ranks cannot run" % _ngpu())
    import hysortk_amd as H
    from hysortk_amd import synth
    cfg = dict(CASES[case])
    R = 2
    if case == "golden_k31":
        seqs = util.read_fasta(util.GOLDEN + "/reads_small.fa")
        cfg["ntasks"] = H.plan_tot_tasks(2, R)
    elif case.startswith("heavy"):
        seqs = _heavy_reads()
    else:
        seqs = list(synth.reads(150000, 150, 12000, 23))
    reads_json = str(tmp_path / "reads.json")
    json.dump(seqs, open(reads_json, "w"))
    spec = dict(cfg, reads=reads_json, out=str(tmp_path / "rank%d.npz"))
    got = _run_ranks(R, spec, tmp_path, 29600 + list(CASES).index(case))
    # the same split through the virtual-rank driver on one GPU
    with H.Context(K=cfg["K"], M=cfg["M"], L=cfg["L"], U=cfg["U"], EXT=cfg["EXT"], ntasks=cfg["ntasks"]) as c:
        want, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in _split(H, seqs, R)])
        st = c.stats()
    for r in range(R):
        g, w = got[r], want[r]
        assert np.array_equal(g["task_off"], w.task_off), (case, r)
        assert np.array_equal(g["kmers"], w.kmers), (case, r)
        assert np.array_equal(g["cnt"], w.cnt), (case, r)
        assert np.array_equal(g["histo"], w.histo), (case, r)
        owned = [t for t in range(len(owner)) if int(g["task_off"][t + 1]) > int(g["task_off"][t])]
        assert all(owner[t] == r for t in owned), (case, r)
        if cfg["EXT"]:
            for i in range(0, len(w), 61):
                a, b = int(g["payload_off"][i]), int(g["payload_off"][i]) + int(g["cnt"][i])
                wp, wr = w.payload(i)
                assert sorted(zip(g["rid"][a:b].tolist(), g["pos"][a:b].tolist())) == sorted(zip(wr.tolist(), wp.tolist())), (case, r, i)
    if case.startswith("heavy"):
        assert st["heavy_tasks"] > 0 and sum(int(g["heavy"][0]) for g in got) > 0
    if case == "golden_k31":
        lines = []
        for g in got:
            lines += ["%s\t%d" % (s, int(c_)) for s, c_ in zip(util.result_strings(g["kmers"], 31), g["cnt"])]
        assert sorted(lines) == open(util.GOLDEN + "/count_k31_np2.txt").read().splitlines()
