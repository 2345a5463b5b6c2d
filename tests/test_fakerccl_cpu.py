"""The stand-in transport of tests/test_gpu_rccl.py (tests/fakerccl) checked by itself, without a GPU: its CPU build moves
host buffers through the same rings.  Message matching in issue order per pair, messages larger than the ring, empty messages,
all-reduce (sum / max, in place), and the two ways it must fail instead of hanging: byte counts that disagree, a peer that
never posts."""
import os
import subprocess
import sys

import pytest

from tests import fakerccl, util


def _run(scenario, nranks, tmp_path, env_extra=None):
    lib = fakerccl.build_cpu()
    idfile = str(tmp_path / ("id_" + scenario))
    env = dict(os.environ, **(env_extra or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(util.ROOT, "tests", "_fakerccl_worker.py"), lib, scenario, str(nranks), str(r), idfile],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=120)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
    return outs


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_exchange_and_allreduce(nranks, tmp_path):
    outs = _run("exchange", nranks, tmp_path)
    assert all(o.strip().endswith("ok") for o in outs), outs


def test_mismatched_byte_counts_fail_on_both_ranks(tmp_path):
    outs = _run("mismatch", 2, tmp_path, {"HSK_FAKERCCL_TIMEOUT": "5"})
    assert all("error:" in o for o in outs), outs


def test_absent_peer_times_out(tmp_path):
    outs = _run("absent", 2, tmp_path, {"HSK_FAKERCCL_TIMEOUT": "1"})
    assert all("error:" in o for o in outs), outs
