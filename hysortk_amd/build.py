"""Builds libhsk.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import fcntl
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
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Several ranks may get here at once (torchrun): one
    of them compiles -- into a temporary file that is renamed over libhsk.so, so nobody ever maps a half-written
    library -- while the others wait on the lock and then find the library up to date."""
    if not force and not needs_build():
        return LIB
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():          # somebody else built it while we waited
                return LIB
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", tmp,
                   os.path.join(SRC_DIR, "hsk_api.hip"), "-ldl"]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def source_sha16():
    """sha256 over the kernel and host sources the library is built from (file names and contents, sorted): the identity of a BUILD that
    profiles/pmc.json and bench.py agree on -- the binary's own hash depends on where and when the compiler ran"""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(SRC_DIR)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode()); h.update(open(os.path.join(SRC_DIR, name), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build(force=True, verbose=True))
