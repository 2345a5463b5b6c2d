"""Builds libhsk.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhsk.so")


def sources():
    return sorted(os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR)) + [os.path.join(os.path.dirname(HERE), "include", "hsk.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB,
           os.path.join(SRC_DIR, "hsk_api.hip"), "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
