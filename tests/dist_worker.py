"""Worker of tests/test_dist_gpu.py: one rank of a multi-process eigenmode run (launched by
torch.distributed.run).  Every rank saves its block's final fields; the test process repeats the run
on the whole mesh with a single block (`run_case(..., None)`) for the bitwise comparison of SURVEY 8(e)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if os.environ.get("SEIGEN_TEST_HANG_DUMP"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SEIGEN_TEST_HANG_DUMP"]), exit=True)
    out, degree, nsteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    n = tuple(int(x) for x in sys.argv[4].split(","))
    import torch
    import torch.distributed as dist
    dev = int(os.environ.get("SEIGEN_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    backend = os.environ.get("SEIGEN_DIST_BACKEND", "gloo")
    if backend == "nccl":      # RCCL, one rank per GPU (tests/test_multigpu_gpu.py)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()

    from seigen_amd.mesh import Partition
    grid = tuple(int(x) for x in sys.argv[5].split(","))
    source = {"source": True, "asym": "asym"}.get(sys.argv[6], False) if len(sys.argv) > 6 else False
    part = Partition(n, rank, world, grid)
    if os.environ.get("SEIGEN_TEST_BAD_PEERS_RANK") == str(rank):
        # tests/test_native_exchange_gpu.py: THIS rank hands the library's argument check a peer on a side that has no
        # neighbour - sg_comm_check refuses it locally, and the ranks must agree to fall back together
        from seigen_amd.backend import HipBlock
        good_check = HipBlock.comm_check

        def bad_check(self, r, nr, peers):
            peers = list(peers)
            peers[[i for i, p in enumerate(peers) if p is None][0]] = (r + 1) % nr
            return good_check(self, r, nr, peers)
        HipBlock.comm_check = bad_check
    el, u, s = run_case(n, degree, nsteps, part, source)
    ex = el._exchanger
    np.savez(os.path.join(out, "rank%d.npz" % rank), u=u, s=s, start=np.array(part.start), n=np.array(part.n),
             bytes_sent=ex.bytes_sent, staged=int(ex.staged), device=dev, native=int(getattr(ex, "native", False)),
             library=str(getattr(ex, "library", "")))
    dist.barrier()
    dist.destroy_process_group()


def case_fields(X):
    """Smooth, asymmetric-looking but symmetric-stress initial data as a function of position."""
    u0 = np.stack([np.sin(3 * X[..., 0]) * np.cos(2 * X[..., 1] + X[..., 2]),
                   np.cos(X[..., 0] - 2 * X[..., 2]), np.sin(X[..., 1] * X[..., 2] + X[..., 0])], axis=-1)
    t = np.cos(2 * X[..., 0] + X[..., 1]) * np.sin(X[..., 2] - X[..., 1])
    s0 = np.zeros(X.shape[:-1] + (3, 3))
    for i in range(3):
        for j in range(3):
            s0[..., i, j] = (1 + i + j) * t + 0.1 * (i + j) * X[..., (i + j) % 3]
    return u0, s0


def run_case(n, degree, nsteps, part, source=False):
    """`nsteps` LF4 steps of the case on the block `part` (None: the whole mesh, no process group).
    source: add a box-Ricker diagonal stress source and a DG4 sponge (3-D explosive source)."""
    import seigen_amd
    from seigen_amd import ElasticLF4, BoxMesh, Expression, Function, FunctionSpace
    import seigen_amd.helpers as helpers
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None
    mesh = BoxMesh(n[0], n[1], n[2], 1.0, 1.0, 1.0)
    if part is not None:
        mesh.set_partition(part)
    el = ElasticLF4.create(mesh, "DG", degree, dimension=3, solver="explicit", output=False)
    el.density, el.mu, el.l = 1.0, 0.25, 0.5
    el.dt = 0.5 * (1.0 / max(n)) / 2 ** (degree - 1)
    u0, s0 = case_fields(el.U.node_coords())      # this block's nodes, global coordinates
    el.u0.dat.data = u0.reshape(-1, 3)
    el.s0.dat.data = s0.reshape(-1, 3, 3)
    if source:
        # the box straddles the block boundaries of every grid the tests use; the sponge too
        box = "x[0] >= 0.2 && x[0] <= 0.7 && x[1] >= 0.3 && x[1] <= 0.8 && x[2] >= 0.3 && x[2] <= 0.8"
        code = "%s ? (-1.0 + 2*a*pow(t - 2.5*dt, 2))*exp(-a*pow(t - 2.5*dt, 2)) : 0.0" % box
        z = "0.0"
        if source == "asym":
            # a NON-symmetric source (xy entry only) that lives in the blocks below z = 0.45 only: those
            # blocks leave symmetric-stress storage on their own, the others must follow (ElasticLF4.
            # _agree_on_stress_storage) or they would mirror ghost traces that are not symmetric
            code = "x[2] <= 0.45 && " + code
            el.source_expression = Expression(((z, code, z), (z, z, z), (z, z, z)), a=4000.0, dt=el.dt, t=0)
        else:
            el.source_expression = Expression(((code, z, z), (z, code, z), (z, z, code)), a=4000.0, dt=el.dt, t=0)
        el.source_function = Function(el.S)
        el.source = el.source_expression
        el.absorption_function = Function(FunctionSpace(mesh, "DG", 4))
        el.absorption = Expression("x[1] >= 0.6 || x[2] <= 0.3 ? 30 : 0")
    # run() as the reference's harness calls it: uploads parameters, sponge and the per-step source table
    el.run(nsteps * el.dt * (1 + 1e-9))
    assert el.block.counters()["steps"] == nsteps
    if source:
        assert el.block.counters()["steps"] > 0 and np.abs(el.source_function.dat.data).max() >= 0
    return el, np.array(el.u1.dat.data_cells), np.array(el.s1.dat.data_cells)


if __name__ == "__main__":
    main()
