"""Worker of tests/test_native_exchange_gpu.py: some ranks of a block grid, each driving the C-ABI DIRECTLY (ctypes; no
torch in this process, so nothing else named nccl* is loaded) with the exchange INSIDE the library (csrc/comm.cpp:
sg_comm_check / sg_comm_init / sg_comm_selftest, then ONE sg_step(n) that runs every stage, pack, grouped
ncclSend / ncclRecv and SECOND launch) over the transport double tests/fake_rccl (SEIGEN_RCCL_LIB names it), all ranks
sharing the test box's one GPU.  The unique id travels through a file, as a host without a process group would do it.

argv: out dir, world, first rank of this process, ranks in this process (threads), grid gx,gy,gz, mesh nx,ny,nz, degree,
steps, dtype, scenario (plain | source | wrap | c3golden)."""
import os
import sys
import threading
import time
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def case_fields(X):
    """smooth initial data as a function of position: every block evaluates it at its own (global) node coordinates"""
    u0 = np.stack([np.sin(3 * X[..., 0]) * np.cos(2 * X[..., 1] + X[..., 2]),
                   np.cos(X[..., 0] - 2 * X[..., 2]), np.sin(X[..., 1] * X[..., 2] + X[..., 0])], axis=-1)
    t = np.cos(2 * X[..., 0] + X[..., 1]) * np.sin(X[..., 2] - X[..., 1])
    s0 = np.zeros(X.shape[:-1] + (3, 3))
    for i in range(3):
        for j in range(3):
            s0[..., i, j] = (1 + i + j) * t + 0.1 * (i + j) * X[..., (i + j) % 3]
    return u0, s0


def sigma_at(Xq):
    """a sponge with the reference's piecewise-constant strips (explosive_source_lf4.py:42-45) on two faces AND a part that
    varies inside the cells (per-cell matrices), straddling the block boundaries of every grid the tests use"""
    strips = np.where((Xq[..., 1] >= 0.625) | (Xq[..., 2] <= 0.25), 30.0, 0.0)
    ramp = np.where(Xq[..., 0] >= 0.5, 40.0 * (Xq[..., 0] - 0.5), 0.0)
    return strips + ramp


def setup_block(blk, n, degree, scenario):
    """parameters, initial fields, source and sponge of the case on `blk` (a block of the n mesh, or the whole of it)"""
    from seigen_amd import _lib
    dt = 0.5 * (1.0 / max(n)) / 2 ** (degree - 1)
    blk.set_params(1.0, dt, 0.5, 0.25)
    u0, s0 = case_fields(blk.node_coords())
    blk.set_field(_lib.FIELD_U, u0)
    blk.set_field(_lib.FIELD_S, s0)
    if scenario == "source":
        blk.set_source_box_ricker([0.2, 0.3, 0.3], [0.7, 0.8, 0.8], 4000.0, 2.5 * dt, dt, dt, 64)
        blk.set_absorption(sigma_at(blk.node_coords(4)), 4)
    return dt


def golden_digest(blk, part, n, gold_cells):
    """What tests/golden/fullsize_c3.npz holds of a field, restricted to this block: the values at the golden's sampled cells
    that lie in the block, and the block's contribution to the sums over every z-layer of cubes of the whole mesh."""
    from seigen_amd import _lib
    ax = [np.arange(part.start[a], part.start[a] + part.n[a]) for a in range(3)]
    cube = (ax[0][None, None, :] + n[0] * (ax[1][None, :, None] + n[1] * ax[2][:, None, None])).reshape(-1)
    gcell = (cube[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)          # global cell of every local cell, local order
    order = np.argsort(gcell)
    pos = np.searchsorted(gcell[order], gold_cells)
    pos[pos >= gcell.size] = 0
    mine = gcell[order][pos] == gold_cells
    local = order[pos[mine]]
    out = {"which": np.nonzero(mine)[0]}
    per_layer = part.n[0] * part.n[1] * 6
    for name, f in (("u", _lib.FIELD_U), ("s", _lib.FIELD_S), ("uh", _lib.FIELD_UH), ("sh", _lib.FIELD_SH)):
        full = blk.get_field(f)
        out[name] = full[local]
        out[name + "_layers"] = full.reshape(part.n[2], per_layer, -1).sum(axis=1)      # local z-layers part.start[2] ..
        del full
    return out


def run_rank(args, rank, failures):
    try:
        _run_rank(args, rank)
    except BaseException:      # noqa: BLE001 - report, and let the other ranks run into the double's timeout
        failures.append("rank %d:\n%s" % (rank, traceback.format_exc()))


def _run_rank(args, rank):
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock, comm_library, comm_unique_id
    from seigen_amd.mesh import Partition
    out, world, grid, n, degree, steps, dtype, scenario = args
    part = Partition(n, rank, world, grid)
    peers = [part.neighbour(s) for s in range(6)]
    mask = part.nbr_mask
    if scenario == "wrap":
        # two blocks around a WRAPPED z axis: both z sides of a block lead to the other rank - two faces between one pair
        # of ranks, which RCCL pairs in posting order (comm.cpp posts the receives in the order of the facing sides)
        assert world == 2 and tuple(grid) == (1, 1, 2)
        peers[4] = peers[5] = 1 - rank
        mask |= 0x30
    h = [1.0 / n[a] for a in range(3)]
    blk = HipBlock(3, degree, part.n, h, [0.0] * 3, "left", mask, dtype=dtype, cube0=list(part.start))
    if scenario == "c3golden":
        # BASELINE config 3's input (the 3-D eigenmode, tests/fullsize_cases.C3) on this rank's block of the 2 x 2 x 2 grid
        import bench
        from seigen_amd import BoxMesh
        from tests import fullsize_cases as fc
        c = fc.C3
        mesh = BoxMesh(n[0], n[1], n[2], 1.0, 1.0, 1.0)
        mesh.set_partition(part)
        blk.set_params(c["rho"], c["dt"], c["lam"], c["mu"])

        class Shim(object):
            pass
        sh = Shim()
        sh.block, sh.mesh, sh.degree = blk, mesh, degree
        bench.fill_initial_condition(sh, c["dt"])
    else:
        setup_block(blk, n, degree, scenario)
    # the unique id: rank 0 makes it, the others find it in the file
    idfile = os.path.join(out, "unique_id.bin")
    if rank == 0:
        with open(idfile + ".tmp", "wb") as f:
            f.write(comm_unique_id())
        os.replace(idfile + ".tmp", idfile)
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise RuntimeError("no unique id from rank 0")
        time.sleep(0.01)
    uid = open(idfile, "rb").read()
    blk.comm_check(rank, world, peers)
    blk.comm_init(uid, rank, world, peers)
    lib, version = comm_library()
    assert os.path.basename(lib) == "libfake_rccl.so", "the exchange is bound to %s, not to the transport double" % lib
    bad = blk.comm_selftest()
    st0 = blk.comm_stats()
    assert st0["exchanges"] == 0 and st0["bytes_sent"] == 0      # the self-test is not part of the run's statistics
    sent = {}
    if scenario == "wrap":
        # one exchange on its own; what each side sent and received is compared ACROSS the ranks by the test
        for field, kind in ((_lib.FIELD_S, 1), (_lib.FIELD_U, 0)):
            blk.comm_exchange(field)
            blk.sync()
            for s in (4, 5):
                sp, rp, nb = blk.comm_buffers(kind, s)
                sent["send_%d_%d" % (kind, s)] = dev_bytes(sp, nb)
                sent["recv_%d_%d" % (kind, s)] = dev_bytes(rp, nb)
        try:
            blk.comm_selftest()
            raise AssertionError("sg_comm_selftest must refuse to overwrite ghost buffers that hold traces")
        except _lib.SeigenHipError:
            pass
        blk.comm_stats(reset=True)
    blk.enable_timing(True)
    blk.step(steps)              # ONE C-ABI call: all stages, packs, exchanges and SECOND launches of all steps
    blk.sync()
    st = blk.comm_stats()
    c = blk.counters()
    nsides = sum(1 for p in peers if p is not None)
    face_bytes = sum(blk.halo_bytes(_lib.FIELD_U, s) for s in range(6) if peers[s] is not None)
    assert st["exchanges"] == 1 + 6 * steps and st["bytes_sent"] == (1 + 6 * steps) * face_bytes, (st, face_bytes)
    assert c["steps"] == steps and all(v == 2 * steps for v in c["launches"])      # FIRST + SECOND of every stage
    if scenario == "c3golden":      # the fields are 1.3 GB per rank: only what the golden holds of them leaves the worker
        gold_cells = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_c3.npz"))["cells"]
        dig = golden_digest(blk, part, n, gold_cells)
        np.savez(os.path.join(out, "rank%d.npz" % rank), start=np.array(part.start), n=np.array(part.n), selftest=bad,
                 exchanges=st["exchanges"], bytes_sent=st["bytes_sent"], nsides=nsides, version=version,
                 **{"g_" + k: v for k, v in dig.items()})
    else:
        np.savez(os.path.join(out, "rank%d.npz" % rank), u=blk.get_field(_lib.FIELD_U), s=blk.get_field(_lib.FIELD_S),
                 uh=blk.get_field(_lib.FIELD_UH), start=np.array(part.start), n=np.array(part.n), selftest=bad,
                 exchanges=st["exchanges"], bytes_sent=st["bytes_sent"], nsides=nsides, version=version, **sent)
    blk.comm_finalize()
    blk.close()


def dev_bytes(ptr, nbytes):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    out = np.empty(nbytes, dtype=np.uint8)
    assert hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), 2) == 0
    return out


def main():
    import faulthandler
    faulthandler.dump_traceback_later(200, exit=True)
    out, world, first, nthreads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    grid = tuple(int(x) for x in sys.argv[5].split(","))
    n = tuple(int(x) for x in sys.argv[6].split(","))
    degree, steps, dtype, scenario = int(sys.argv[7]), int(sys.argv[8]), sys.argv[9], sys.argv[10]
    assert "torch" not in sys.modules
    args = (out, world, grid, n, degree, steps, dtype, scenario)
    failures = []
    threads = [threading.Thread(target=run_rank, args=(args, r, failures)) for r in range(first, first + nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert "torch" not in sys.modules, "the worker must stay free of torch (and of its RCCL)"
    if failures:
        sys.stderr.write("\n".join(failures))
        sys.exit(1)


if __name__ == "__main__":
    main()
