"""Worker of tests/test_dist_gpu.py::test_rccl_transport_on_one_rank: the RCCL transport of the halo
layer (seigen_amd/parallel.py, backend "nccl") exercised on ONE GPU.  RCCL refuses two ranks on one
device, so the process group has a single rank whose block is its own neighbour across z (a send and a
receive to oneself inside one grouped call are legal): every torch.distributed call of the multi-GPU
path runs for real - group init with a device id, batch_isend_irecv of device buffers issued against
the library's launch stream, the work handles' wait, the event pair around it - and what arrives can be
checked exactly: the traces received on side z- are the ones packed on side z+ and vice versa."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import torch.distributed as dist
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    from seigen_amd.mesh import Partition
    from seigen_amd.parallel import HaloExchanger

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"

    class SelfNeighbour(Partition):
        def neighbour(self, side):
            return 0 if side >> 1 == 2 else None          # z- and z+ lead back to this rank

    for dtype in ("f64", "f32"):
        n, P = (16, 4, 6), 3
        part = SelfNeighbour(n, 0, 1)
        blk = HipBlock(3, P, n, [1.0 / 16] * 3, [0.0] * 3, "left", part.nbr_mask, dtype=dtype)
        assert blk.nbr_mask == 0x30
        stream = torch.cuda.ExternalStream(blk.stream_ptr(), device=0)
        ex = HaloExchanger(blk, part, torch.device("cuda", 0), stream=stream)
        assert not ex.staged and ex.sides == [4, 5]
        rng = np.random.default_rng(0)
        u0 = rng.uniform(-1, 1, blk.field_shape(_lib.FIELD_U))
        s0 = rng.uniform(-1, 1, blk.field_shape(_lib.FIELD_S))
        s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        blk.set_params(1.0, 1e-4, 0.5, 0.25)
        blk.set_field(_lib.FIELD_U, u0)
        blk.set_field(_lib.FIELD_S, s0)
        ex.reset_stats(timing=True)
        for field, kind in ((_lib.FIELD_S, "s"), (_lib.FIELD_U, "u")):
            ex.finish(ex.start(field))
            blk.sync()
            torch.cuda.synchronize()
            for s in (4, 5):
                # both sides talk to the same peer (this rank): RCCL pairs the sends and receives of one peer in posting
                # order, and the exchanger posts its receives in the order of the FACING sides - side s receives what
                # was packed for side s ^ 1 (the block is its own periodic image across z)
                sent, got = ex.send[(kind, s ^ 1)], ex.recv[(kind, s)]
                assert float(sent.abs().max()) > 0
                assert torch.equal(sent, got), "side %d did not receive the trace of the facing side" % s
            assert not torch.equal(ex.recv[(kind, 4)], ex.recv[(kind, 5)])
        ex.step(3)                                         # the pipelined schedule, three whole steps
        blk.sync()
        torch.cuda.synchronize()
        st = ex.stats()
        assert st["exchanges"] == 2 + 1 + 18 and st["exposed_wait_ms"] >= 0.0
        u = blk.get_field(_lib.FIELD_U)
        assert np.isfinite(u).all() and np.abs(u - u0).max() > 0
        print("rccl self-exchange ok (%s): %d exchanges, %.3f ms waited on the stream, %d bytes sent"
              % (dtype, st["exchanges"], st["exposed_wait_ms"], st["bytes_sent"]))
        blk.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
