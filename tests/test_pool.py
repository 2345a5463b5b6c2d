"""The device pool's segment logic (hysortk_amd/csrc/hsk_pool.h: best fit, split, coalesce, trim) against malloc on the CPU, under the
address and undefined-behaviour sanitizers: tests/pool_test.cpp allocates, patterns, checks and releases a few hundred thousand blocks."""
import os
import subprocess

from tests import util


def test_pool_segments_under_sanitizers(tmp_path):
    exe = str(tmp_path / "pool_test")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17",
                           os.path.join(util.ROOT, "tests", "pool_test.cpp"), "-o", exe])
    for seed in (1, 7):
        out = subprocess.check_output([exe, str(seed)]).decode()
        assert out.startswith("OK"), out
