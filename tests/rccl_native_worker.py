"""Worker of tests/test_dist_gpu.py::test_native_rccl_exchange_on_one_rank: the exchange INSIDE the library
(csrc/comm.cpp: sg_comm_init + sg_step) on ONE GPU.  RCCL refuses two ranks on one device, so the communicator has a
single rank whose block is its own neighbour across z (a send and a receive to oneself inside one grouped call are
legal): ncclCommInitRank, the grouped ncclSend / ncclRecv on the handle's stream, the two-stream schedule all run for
real.  Checked: what arrives is what was packed, and the native run equals - bit for bit - the run of the host-driven
exchanger (seigen_amd/parallel.py over torch.distributed's RCCL group) on an identical block, double and float."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dev_bytes(ptr, nbytes):
    hip = ctypes.CDLL("libamdhip64.so")
    out = np.empty(nbytes, dtype=np.uint8)
    assert hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
    return out


def main():
    import torch
    import torch.distributed as dist
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock, comm_unique_id
    from seigen_amd.mesh import Partition
    from seigen_amd.parallel import HaloExchanger

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))

    class SelfNeighbour(Partition):
        def neighbour(self, side):
            return 0 if side >> 1 == 2 else None          # z- and z+ lead back to this rank

    for dtype in ("f64", "f32"):
        n, P = (16, 4, 6), 3
        part = SelfNeighbour(n, 0, 1)
        rng = np.random.default_rng(0)

        def fresh():
            b = HipBlock(3, P, n, [1.0 / 16] * 3, [0.0] * 3, "left", part.nbr_mask, dtype=dtype)
            b.set_params(1.0, 1e-4, 0.5, 0.25)
            return b

        nat, ref = fresh(), fresh()
        u0 = rng.uniform(-1, 1, nat.field_shape(_lib.FIELD_U))
        s0 = rng.uniform(-1, 1, nat.field_shape(_lib.FIELD_S))
        s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        for b in (nat, ref):
            b.set_field(_lib.FIELD_U, u0)
            b.set_field(_lib.FIELD_S, s0)
        # a source whose nodes lie in the shell and in the interior, and a sponge
        nodes = np.unique(rng.integers(0, nat.ncells * nat.nd, size=40))
        sv = rng.uniform(-1, 1, size=(8, len(nodes), 3, 3))
        sv = 0.5 * (sv + np.swapaxes(sv, -1, -2))
        sigma = np.where(rng.uniform(size=(nat.ncells, 35)) > 0.7, 2.0, 0.0)
        for b in (nat, ref):
            b.set_source(nodes, sv)
            b.set_absorption(sigma, 4)
        nat.comm_init(comm_unique_id(), 0, 1, [None, None, None, None, 0, 0])
        # which RCCL the library bound (torch's bundled copy or the system's): a major version 2 copy with a name
        from seigen_amd.backend import comm_library
        lib_path, lib_version = comm_library()
        assert lib_path and lib_version // 10000 == 2, (lib_path, lib_version)
        print("native exchange bound to %s (version %d)" % (lib_path, lib_version))
        # the pattern self-test of the exchange (what NativeExchanger runs before it trusts the communicator)
        assert nat.comm_selftest() == 0
        assert nat.comm_stats()["exchanges"] == 0 and nat.comm_stats()["bytes_sent"] == 0
        # one exchange on its own: a side receives what the FACING side of its neighbour packed - here, the block being
        # its own neighbour across z, side 5 gets the trace of side 4 and vice versa (receives are posted in the order
        # of the facing sides: RCCL pairs the sends and receives of one peer in posting order) - and the two differ
        for field, kind in ((_lib.FIELD_S, 1), (_lib.FIELD_U, 0)):
            nat.comm_exchange(field)
            nat.sync()
            sent, got = {}, {}
            for s in (4, 5):
                sp, rp, nb = nat.comm_buffers(kind, s)
                sent[s], got[s] = dev_bytes(sp, nb), dev_bytes(rp, nb)
                assert nb == nat.halo_bytes(field, s) and sent[s].any()
            for s in (4, 5):
                assert np.array_equal(got[s], sent[s ^ 1]), "side %d did not receive the trace of the facing side" % s
            assert not np.array_equal(got[4], got[5])
        assert nat.comm_stats(reset=True)["exchanges"] == 2
        # whole steps: one C-ABI call, against the host-driven exchanger on the twin block
        nat.enable_timing(True)
        nat.step(3)
        nat.sync()
        st = nat.comm_stats()
        assert st["exchanges"] == 1 + 18 and st["bytes_sent"] == 19 * 2 * nat.halo_bytes(_lib.FIELD_U, 4)
        assert st["exposed_wait_ms"] >= 0.0
        c = nat.counters()
        assert c["steps"] == 3 and all(v == 6 for v in c["launches"])      # FIRST + SECOND of every stage
        stream = torch.cuda.ExternalStream(ref.stream_ptr(), device=0)
        ex = HaloExchanger(ref, part, torch.device("cuda", 0), stream=stream)
        ex.step(3)
        ref.sync()
        torch.cuda.synchronize()
        for f in (_lib.FIELD_U, _lib.FIELD_S, _lib.FIELD_UH, _lib.FIELD_SH):
            a, b = nat.get_field(f), ref.get_field(f)
            assert np.isfinite(a).all() and np.array_equal(a, b), "native and host-driven exchange differ (field %d)" % f
        assert np.abs(nat.get_field(_lib.FIELD_U) - u0).max() > 0
        # a second call continues the run (the first input's halo is exchanged again up front)
        nat.step(2)
        ex.step(2)
        nat.sync()
        ref.sync()
        torch.cuda.synchronize()
        assert np.array_equal(nat.get_field(_lib.FIELD_S), ref.get_field(_lib.FIELD_S))
        # the host-side wrapper the solver class uses (unique id from rank 0 through the process group's broadcast,
        # peers from the partition, statistics): a third twin block stepped through it
        from seigen_amd.parallel import NativeExchanger
        third = fresh()
        third.set_field(_lib.FIELD_U, u0)
        third.set_field(_lib.FIELD_S, s0)
        third.set_source(nodes, sv)
        third.set_absorption(sigma, 4)
        nex = NativeExchanger(third, part)
        assert nex.sides == [4, 5] and not nex.staged and nex.native
        nex.reset_stats(timing=True)
        nex.step(3)
        nex.step(2)
        third.sync()
        assert np.array_equal(third.get_field(_lib.FIELD_S), ref.get_field(_lib.FIELD_S))
        st3 = nex.stats()
        assert st3["exchanges"] == 2 + 30 and nex.bytes_sent == st3["bytes_sent"] > 0 and st3["host_blocked_ms"] == 0.0
        third.close()
        print("native rccl exchange ok (%s): %d exchanges, %.3f ms waited beyond SECOND, %d bytes sent"
              % (dtype, st["exchanges"], st["exposed_wait_ms"], st["bytes_sent"]))
        nat.close()
        ref.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
