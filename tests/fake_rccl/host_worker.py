"""One process of tests/test_fake_rccl.py: `nthreads` ranks of a fake_rccl communicator between HOST buffers
(FAKE_RCCL_HOST=1), driven through ctypes exactly as csrc/comm.cpp drives RCCL.  argv: library, id file, world size,
first rank of this process, ranks in this process, scenario."""
import ctypes as C
import json
import sys
import threading

import numpy as np


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def bind(path):
    lib = C.CDLL(path)
    lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    return lib


NCCL_DOUBLE = 8   # ncclFloat64 (rccl.h)


def payload(src, dst, k, n):
    return np.arange(n, dtype=np.float64) + 1e6 * src + 1e4 * dst + 1e2 * k


def run_rank(lib, uid, world, rank, scenario, out):
    comm = C.c_void_p()
    rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
    if rc != 0:
        out[rank] = {"init": rc}
        return
    res = {"init": 0, "rounds": []}
    n = 1000
    for rnd in range(3):
        # to every peer (myself included): two messages of different content; from every peer: two receives, posted in
        # the opposite peer order - only the posting order WITHIN a pair of ranks may matter
        sends = {(p, k): payload(rank, p, 10 * rnd + k, n + p + k) for p in range(world) for k in range(2)}
        recvs = {(p, k): np.full(n + rank + k, -1.0) for p in range(world) for k in range(2)}
        if scenario == "mismatch" and rank == 1 and rnd == 1:
            recvs[(0, 0)] = np.full(7, -1.0)          # a receive that does not match its send
        rcs = [lib.ncclGroupStart()]
        for p in range(world):
            for k in range(2):
                a = sends[(p, k)]
                rcs.append(lib.ncclSend(a.ctypes.data, a.size, NCCL_DOUBLE, p, comm, None))
        for p in reversed(range(world)):
            for k in range(2):
                b = recvs[(p, k)]
                rcs.append(lib.ncclRecv(b.ctypes.data, b.size, NCCL_DOUBLE, p, comm, None))
        rcs.append(lib.ncclGroupEnd())
        ok = all(r == 0 for r in rcs)
        good = ok and all(np.array_equal(recvs[(p, k)], payload(p, rank, 10 * rnd + k, n + rank + k))
                          for p in range(world) for k in range(2))
        res["rounds"].append({"rc": max(rcs), "good": bool(good)})
        if not ok:
            break
    lib.ncclCommDestroy(comm)
    out[rank] = res


def main():
    path, idfile, world, first, nthreads, scenario = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    lib = bind(path)
    uid = UniqueId()
    raw = open(idfile, "rb").read()
    C.memmove(C.byref(uid), raw, 128)
    out = {}
    ranks = [r for r in range(first, first + nthreads) if not (scenario == "absent" and r == world - 1)]
    threads = [threading.Thread(target=run_rank, args=(lib, uid, world, r, scenario, out)) for r in ranks]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print(json.dumps({str(k): v for k, v in out.items()}))


if __name__ == "__main__":
    main()
