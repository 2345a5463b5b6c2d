"""Worker of tests/test_fakerccl_cpu.py: one rank of the stand-in transport's CPU build (tests/fakerccl, -DFAKERCCL_NO_HIP:
"device" buffers are host buffers).  argv: library, scenario, nranks, rank, id-file."""
import ctypes as C
import os
import sys
import time

import numpy as np


class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def main():
    lib_path, scenario, nranks, rank, idfile = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    L = C.CDLL(lib_path)
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    L.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclGetErrorString.restype = C.c_char_p
    uid = Uid()
    if rank == 0:
        assert L.ncclGetUniqueId(C.byref(uid)) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.replace(idfile + ".tmp", idfile)
    else:
        while not os.path.exists(idfile):
            time.sleep(0.01)
        C.memmove(C.byref(uid), open(idfile, "rb").read(), 128)
    comm = C.c_void_p()
    rc = L.ncclCommInitRank(C.byref(comm), nranks, uid, rank)
    assert rc == 0, L.ncclGetErrorString(rc)

    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    def pattern(src, dst, n, salt=0):
        return ((np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(src * 1000003 + dst * 101 + salt)) & np.uint64(0xFF)).astype(np.uint8)

    def msg_bytes(src, dst, j):           # sizes around and beyond the ring (2 slots x 4 MB), one empty, one tiny
        return [0, 7, (9 << 20) + 13 * src + dst, 4 << 20][(src + dst + j) % 4]

    if scenario == "exchange":
        v = np.array([rank + 1, 10 * (rank + 1), 7], dtype=np.uint64)
        out = np.zeros_like(v)
        assert L.ncclAllReduce(ptr(v), ptr(out), 3, 5, 0, comm, None) == 0
        assert out.tolist() == [sum(r + 1 for r in range(nranks)), sum(10 * (r + 1) for r in range(nranks)), 7 * nranks], out
        assert L.ncclAllReduce(ptr(v), ptr(v), 3, 5, 2, comm, None) == 0          # in place, max
        assert v.tolist() == [nranks, 10 * nranks, 7], v
        for rnd in range(3):                                  # several groups back to back: two messages per peer and direction
            sends, recvs = [], []
            assert L.ncclGroupStart() == 0
            for q in range(nranks):
                for j in range(2):
                    s = pattern(rank, q, msg_bytes(rank, q, j + rnd), j)
                    r = np.zeros(msg_bytes(q, rank, j + rnd), dtype=np.uint8)
                    sends.append(s); recvs.append((q, j, r))
                    assert L.ncclSend(ptr(s), s.size, 1, q, comm, None) == 0
                    assert L.ncclRecv(ptr(r), r.size, 1, q, comm, None) == 0
            rc = L.ncclGroupEnd()
            assert rc == 0, L.ncclGetErrorString(rc)
            for q, j, r in recvs:
                assert np.array_equal(r, pattern(q, rank, r.size, j)), (rnd, q, j)
        print("ok")
    elif scenario == "mismatch":                               # byte counts that disagree: an error on both sides, not a hang
        n = 100 if rank == 0 else 99
        s, r = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        other = 1 - rank
        L.ncclGroupStart()
        L.ncclSend(ptr(s), n, 1, other, comm, None)
        L.ncclRecv(ptr(r), n, 1, other, comm, None)
        rc = L.ncclGroupEnd()
        assert rc != 0
        print("error:", L.ncclGetErrorString(rc).decode())
    elif scenario == "absent":                                 # rank 1 never posts: rank 0 times out instead of hanging
        if rank == 0:
            r = np.zeros(10, dtype=np.uint8)
            t0 = time.time()
            rc = L.ncclRecv(ptr(r), 10, 1, 1, comm, None)
            assert rc != 0 and time.time() - t0 < 30
            v = np.zeros(1, dtype=np.uint64)
            assert L.ncclAllReduce(ptr(v), ptr(v), 1, 5, 0, comm, None) != 0      # the communicator stays aborted
            print("error:", L.ncclGetErrorString(rc).decode())
        else:
            time.sleep(3)
            v = np.zeros(1, dtype=np.uint64)
            assert L.ncclAllReduce(ptr(v), ptr(v), 1, 5, 0, comm, None) != 0
            print("error: aborted")
    L.ncclCommDestroy(comm)


if __name__ == "__main__":
    main()
