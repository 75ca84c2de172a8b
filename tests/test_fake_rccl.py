"""The transport double of the native halo exchange's tests (tests/fake_rccl/fake_rccl.c) checked ON ITS OWN, between
host buffers (FAKE_RCCL_HOST=1; no GPU): does it keep the RCCL semantics csrc/comm.cpp relies on - sends and receives
between a pair of ranks pair up in posting order, a group may address several peers and the rank itself - and does it
turn the situations that would hang real RCCL (a rank that never arrives, a receive that matches the wrong send) into
errors on every rank.  The GPU tests that run the product's exchange over it are tests/test_native_exchange_gpu.py."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fake_rccl.build import build  # noqa: E402
from fake_rccl.host_worker import UniqueId, bind  # noqa: E402


@pytest.fixture(scope="module")
def fake():
    return build()


def _run(fake, tmp_path, world, layout, scenario="plain", timeout_s="20"):
    """layout: ranks per process, e.g. [1, 1] = two processes, [2, 2] = two processes with two ranks (threads) each"""
    lib = bind(fake)
    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    idfile = tmp_path / "uid.bin"
    idfile.write_bytes(bytes(uid))
    env = dict(os.environ, FAKE_RCCL_HOST="1", FAKE_RCCL_TIMEOUT_S=timeout_s, FAKE_RCCL_SLOT_BYTES="65536",
               FAKE_RCCL_LOG=str(tmp_path / "log"))
    procs, first = [], 0
    for n in layout:
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "fake_rccl", "host_worker.py"), fake,
                                       str(idfile), str(world), str(first), str(n), scenario],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        first += n
    out = {}
    t0 = time.time()
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se[-2000:]
        out.update(json.loads(so.strip().splitlines()[-1]))
    name = uid.internal.decode().split("/", 1)[1]
    assert not os.path.exists("/dev/shm/" + name), "the shared-memory segment must not outlive the communicator"
    return out, time.time() - t0


@pytest.mark.parametrize("world,layout", [(2, [1, 1]), (3, [1, 1, 1]), (4, [2, 2]), (8, [2, 2, 2, 2]), (1, [1])])
def test_pairs_in_posting_order(fake, tmp_path, world, layout):
    out, _ = _run(fake, tmp_path, world, layout)
    assert sorted(out) == sorted(str(r) for r in range(world))
    for r, res in out.items():
        assert res["init"] == 0 and len(res["rounds"]) == 3 and all(x["rc"] == 0 and x["good"] for x in res["rounds"]), (r, res)
    for r in range(world):
        st = json.loads(open(str(tmp_path / "log") + ".rank%d" % r).read())
        assert st["sends"] == st["recvs"] == 3 * 2 * world and st["host_mode"] == 1 and st["groups"] == 3


def test_a_rank_that_never_arrives_is_an_error_not_a_hang(fake, tmp_path):
    out, took = _run(fake, tmp_path, 3, [1, 1, 1], scenario="absent", timeout_s="2")
    assert sorted(out) == ["0", "1"] and all(res["init"] != 0 for res in out.values())
    assert took < 30


def test_a_mismatched_receive_poisons_every_rank(fake, tmp_path):
    out, took = _run(fake, tmp_path, 2, [1, 1], scenario="mismatch", timeout_s="5")
    assert out["1"]["rounds"][1]["rc"] != 0 and not out["1"]["rounds"][1]["good"]
    # rank 0 finishes round 1 or fails in it, and cannot complete a later round: the communicator is poisoned
    assert not all(x["rc"] == 0 for x in out["0"]["rounds"]) or len(out["0"]["rounds"]) < 3
    assert took < 30


def test_ids_from_elsewhere_and_oversized_messages_are_refused(fake, tmp_path):
    lib = bind(fake)
    uid = UniqueId()
    comm = C.c_void_p()
    assert lib.ncclCommInitRank(C.byref(comm), 1, uid, 0) != 0      # all-zero id: not made by the double
    os.environ["FAKE_RCCL_HOST"] = "1"
    os.environ["FAKE_RCCL_SLOT_BYTES"] = "4096"
    try:
        assert lib.ncclGetUniqueId(C.byref(uid)) == 0
        assert lib.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
        buf = (C.c_double * 1024)()
        assert lib.ncclSend(buf, 1024, 8, 0, comm, None) != 0       # 8 KB into 4 KB slots
        assert lib.ncclSend(buf, 512, 8, 3, comm, None) != 0        # no such peer
        assert lib.ncclGroupEnd() != 0                              # no group open
        assert lib.ncclSend(buf, 512, 8, 0, comm, None) == 0 and lib.ncclRecv(buf, 512, 8, 0, comm, None) == 0
        assert lib.ncclCommDestroy(comm) == 0
        v = C.c_int()
        assert lib.ncclGetVersion(C.byref(v)) == 0 and v.value // 10000 == 2
    finally:
        os.environ.pop("FAKE_RCCL_HOST")
        os.environ.pop("FAKE_RCCL_SLOT_BYTES")
