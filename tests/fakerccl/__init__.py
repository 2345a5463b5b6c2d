"""TEST-ONLY stand-in for librccl (see fakerccl.cpp): builders for its two variants.

libfakerccl.so      links libamdhip64: N processes on ONE GPU exchange device buffers (HSK_RCCL_LIB points hsk_comm.h at it)
libfakerccl_cpu.so  -DFAKERCCL_NO_HIP: host buffers only, for the protocol tests that run without a GPU
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fakerccl.cpp")
LIB = os.path.join(HERE, "libfakerccl.so")
LIB_CPU = os.path.join(HERE, "libfakerccl_cpu.so")


def _stale(lib):
    return not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(SRC)


def build(force=False):
    if force or _stale(LIB):
        tmp = "%s.tmp.%d" % (LIB, os.getpid())
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", tmp, SRC, "-lrt"])
        os.replace(tmp, LIB)
    return LIB


def build_cpu(force=False):
    if force or _stale(LIB_CPU):
        tmp = "%s.tmp.%d" % (LIB_CPU, os.getpid())
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-DFAKERCCL_NO_HIP", "-o", tmp, SRC, "-lrt"])
        os.replace(tmp, LIB_CPU)
    return LIB_CPU
