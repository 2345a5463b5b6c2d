"""The real N>1 path: two fresh processes form a communicator (hsk_comm_init) and run hsk_count on their halves of the reads
-- grouped ncclSend/ncclRecv per task group on the second stream (post_exchange, csrc/hsk_comm.h), all-reduces of task sizes
/ size matrix with the ranks' status, heavy-hitter list exchange, leaving together after a failure.  Per-rank lists must equal
what the virtual-rank driver (hsk_count_loopback, device copies in place of the transport) gives on one GPU, and their union
the reference's 2-rank output.

With two GPUs the ranks take one each and RCCL carries the data.  On a ONE-GPU box (RCCL refuses two ranks on one device) both
ranks open the same GPU (HSK_FORCE_DEVICE=0) and HSK_RCCL_LIB points csrc/hsk_comm.h at tests/fakerccl, a stand-in for the
nine entry points it binds that moves the messages through shared memory: the product code that runs is the same."""
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


def _transport_env():
    """{} with two GPUs (RCCL); the stand-in transport and a shared device otherwise."""
    if _ngpu() >= 2 and not os.environ.get("HSK_TEST_FORCE_FAKERCCL"):
        return {}
    from tests import fakerccl
    return {"HSK_RCCL_LIB": fakerccl.build(), "HSK_FORCE_DEVICE": "0", "HSK_FAKERCCL_TIMEOUT": "90"}


def _run_ranks(world, spec, tmp_path, port, extra_env=None):
    procs = []
    tenv = _transport_env()
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **tenv)
        env.update(util.tune_env(extra_env))
        procs.append(subprocess.Popen([sys.executable, os.path.join(util.ROOT, "tests", "_rccl_worker.py"), json.dumps(spec)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=420)[0].decode())
        except subprocess.TimeoutExpired:                  # the watchdog: a hang of the ranks fails the test, the children are killed by pid
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
    assert sum(int(g["total_kmers"][0]) for g in got) == sum(len(s_) - cfg["K"] + 1 for s_ in seqs if len(s_) >= cfg["K"])      # (heavy tasks' instances included)
    if case == "golden_k31":
        lines = []
        for g in got:
            lines += ["%s\t%d" % (s, int(c_)) for s, c_ in zip(util.result_strings(g["kmers"], 31), g["cnt"])]
        assert sorted(lines) == open(util.GOLDEN + "/count_k31_np2.txt").read().splitlines()


def _small_case(tmp_path):
    from hysortk_amd import synth
    seqs = list(synth.reads(150000, 150, 12000, 23))
    reads_json = str(tmp_path / "reads.json")
    json.dump(seqs, open(reads_json, "w"))
    return seqs, reads_json


@pytest.mark.parametrize("site", ["sortbuf", "group1", "late"])
def test_failing_together(site, tmp_path):
    """One rank's allocation fails -- before the first task group travels (sortbuf), after it (group1: the exchange buffers of
    the second group) or in the middle of the batches (late).  BOTH ranks must return an error, quickly, nobody hangs, and the
    same contexts and communicator count correctly afterwards (reference: the ranks die together, src/kmerops.cpp:1477)."""
    import hysortk_amd as H
    cfg = dict(K=31, M=17, L=2, U=50, EXT=0, ntasks=40)          # 20 tasks per rank: three task groups, three batches
    seqs, reads_json = _small_case(tmp_path)
    spec = dict(cfg, reads=reads_json, out=str(tmp_path / "rank%d.npz"), fail="1:" + site)
    got = _run_ranks(2, spec, tmp_path, 29650 + ["sortbuf", "group1", "late"].index(site))
    assert int(got[1]["fail_code"][0]) == 4, str(got[1]["fail_msg"])                                   # HSK_ERR_OOM: its own error
    assert "injected" in str(got[1]["fail_msg"][0])
    assert int(got[0]["fail_code"][0]) == 7 and "another rank" in str(got[0]["fail_msg"][0]), str(got[0]["fail_msg"])     # HSK_ERR_COMM
    assert float(got[0]["fail_seconds"][0]) < 60 and float(got[1]["fail_seconds"][0]) < 60
    with H.Context(**cfg) as c:
        want, _ = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in _split(H, seqs, 2)])
    for r in range(2):
        assert np.array_equal(got[r]["task_off"], want[r].task_off) and np.array_equal(got[r]["kmers"], want[r].kmers) and np.array_equal(got[r]["cnt"], want[r].cnt), (site, r)


def test_long_messages_travel_in_pieces(tmp_path):
    """HSK_RCCL_MSG_MAX: every send / receive longer than 64 KB is cut into several (what 80 Gbp without the group overlap needs
    beyond 2^31 bytes per message); same lists."""
    import hysortk_amd as H
    cfg = dict(K=31, M=17, L=2, U=50, EXT=1, ntasks=16)
    seqs, reads_json = _small_case(tmp_path)
    spec = dict(cfg, reads=reads_json, out=str(tmp_path / "rank%d.npz"))
    for overlap in ("1", "0"):
        got = _run_ranks(2, spec, tmp_path, 29660 + int(overlap), {"HSK_RCCL_MSG_MAX": "65536", "HSK_OVERLAP": overlap})
        with H.Context(**cfg) as c:
            want, _ = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in _split(H, seqs, 2)])
        for r in range(2):
            assert np.array_equal(got[r]["kmers"], want[r].kmers) and np.array_equal(got[r]["cnt"], want[r].cnt), (overlap, r)


def test_bench_launcher_two_ranks(tmp_path):
    """`python bench.py --gpus 2` starts its ranks itself (torch.distributed.run child) and prints one JSON line with n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **_transport_env())
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--scale", "0.01", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    js = json.loads(line)
    assert js["n_gpus"] == 2 and js["value"] > 0 and js["config"]["exchange"].startswith("RCCL")
    assert js["phases_ms_per_step"]["ms_exchange"] >= 0


@pytest.mark.parametrize("world", [2, 4])
def test_two_ranks_owner_side_combining_extraction(tmp_path, world):
    """The combining extraction with several ranks, over the real exchange code: the supermers travel with 16 of their minimizer bits
    (a third array beside len[] and bytes[] in every grouped send / receive), every owner builds the items of its tasks and counts them per
    minimizer bucket.  Both ranks' lists equal the instance path's (virtual ranks in this process, library defaults: inputs of this size stay
    on the instance path)."""
    import hysortk_amd as H
    from hysortk_amd import synth
    if world > 2 and _ngpu() >= 2 and _ngpu() < world:
        pytest.skip("four ranks: four GPUs, or one GPU shared through the stand-in transport")
    cfg = dict(K=31, M=17, L=1, U=65535, EXT=0, ntasks=24 * world)     # 24 tasks per rank: three task groups each
    seqs = list(synth.reads(300000, 150, 40000, 29))
    reads_json = str(tmp_path / "reads.json")
    json.dump(seqs, open(reads_json, "w"))
    spec = dict(cfg, reads=reads_json, out=str(tmp_path / "rank%d.npz"))
    got = _run_ranks(world, spec, tmp_path, 29670 + world, {"HSK_COMBINE_MIN_BYTES": "0"})
    with H.Context(**cfg) as c:
        want, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in _split(H, seqs, world)])
        st = c.stats()
    assert st["combine_launches"] == 0
    for r in range(world):
        assert int(got[r]["combine_pairs"][0]) > 0, r                       # (pairs were written: combine_kernel ran on this rank's tasks)
        assert np.array_equal(got[r]["task_off"], want[r].task_off) and np.array_equal(got[r]["kmers"], want[r].kmers) and np.array_equal(got[r]["cnt"], want[r].cnt), r


def test_two_ranks_agree_on_the_certain_drops(tmp_path):
    """Only rank 0's reads hold the all-A reads (700 x 120 copies of the all-A k-mer: more than U and more than 2^16 inside its sketch's sample);
    rank 1's sketch sees none.  The masks are ORed in the plan's all-reduce (run_pipeline), so rank 1 -- which owns the all-A k-mer's task or
    not -- leaves them out as well; the lists equal the ones without the drops, total_kmers still counts the instances."""
    import hysortk_amd as H
    from hysortk_amd import synth
    cfg = dict(K=31, M=17, L=1, U=65535, EXT=0, ntasks=48)
    seqs = ["A" * 150] * 700 + list(synth.reads(300000, 150, 40000, 29))
    reads_json = str(tmp_path / "reads.json")
    json.dump(seqs, open(reads_json, "w"))
    spec = dict(cfg, reads=reads_json, out=str(tmp_path / "rank%d.npz"))
    got = _run_ranks(2, spec, tmp_path, 29690, {"HSK_PLAN_MIN_INPUT": "1"})
    with H.Context(tuning="drop_certain=0", **cfg) as c:
        want, owner = c.count_loopback([H.DnaBuffer.from_sequences(p) for p in _split(H, seqs, 2)])
    assert int(got[0]["dropped"][0]) == 700 * 120 and int(got[1]["dropped"][0]) == 0
    # (the all-A task is smaller without its 84 000 instances, so the dispatcher may hand the tasks out differently: task by task, whoever owns it)
    def task_list(lists, t, get):
        for x in lists:
            to = get(x, "task_off")
            a, b = int(to[t]), int(to[t + 1])
            if b > a:
                return get(x, "kmers")[a:b], get(x, "cnt")[a:b]
        return None
    n = 0
    for t in range(len(owner)):
        g, w = task_list(got, t, lambda x, k: x[k]), task_list(want, t, lambda x, k: getattr(x, k))
        assert (g is None) == (w is None), t
        if g is not None:
            assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), t
            n += len(g[1])
    assert n == sum(len(w) for w in want) > 100000
    assert sum(int(g["total_kmers"][0]) for g in got) == sum(int(w.info["total_kmers"]) for w in want) == len(seqs) * 120
