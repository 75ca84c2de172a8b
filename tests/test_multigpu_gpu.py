"""More than one MI355X: the tests that switch themselves on when the box has them.

The reference's protocol runs NP in {1, 2, 4, 8, 16} MPI ranks (tests/eigenmode/README.md:7-13) with the halo
exchange implicit in every assemble (seigen/elastic.py:364, :404-436).  Here: one process per GPU, RCCL send/receive
of the packed facet traces (`T.n`) between face neighbours.  With >= 2 visible devices these tests start 2 ranks
(>= 4 devices: 4 - the box's process guard allows six GPU processes, so the eight ranks of the 2 x 2 x 2 grid are left to
the driver's scaling run) with the **nccl** backend, through the solver class and through bench.py, and require
  * every rank's block bitwise equal to the single-block run of the whole mesh,
  * `n_gpus == N`, `halo.transport == "nccl"`, and the bytes sent per step equal to the closed-form `T.n` payload.
On a one-GPU box they SKIP (they do not pass); what a one-GPU box can check of config 4's partition is the last
test: config 3's golden input split 2 x 2 x 2 into eight 32^3 blocks on one device (device-copy transport) against the
oracle's full-size golden (tests/golden/fullsize_c3.npz)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ndev():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
    except OSError:
        return 0


def _nranks():
    """Ranks to start: one per device, but never more than four - a GPU box allows six processes on its cards at once,
    and the test process itself is one of them (the 2 x 2 x 2 grid of eight ranks is the driver's scaling run)."""
    n = _ndev()
    return 4 if n >= 4 else (2 if n >= 2 else 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch_nccl(nproc, script_args, timeout=600):
    env = dict(os.environ, SEIGEN_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               SEIGEN_TEST_HANG_DUMP="400")
    env.pop("SEIGEN_HIP_DEVICE", None)      # rank r drives GPU r (LOCAL_RANK)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


# mesh and process grid per rank count: blocks wider than 32 cubes along x wherever x is cut (the shell next to an x
# side is a whole 16-cube layout group thick, csrc/handle.hpp shell_width_x)
CASES = {2: ((16, 4, 8), (1, 1, 2)), 4: ((80, 8, 4), (2, 2, 1))}      # four ranks: an x cut and a y cut


@pytest.mark.parametrize("degree,source", [(4, False), (4, True), (3, "asym")])
def test_nccl_ranks_equal_the_single_block_bitwise(gpu, tmp_path, degree, source):
    world = _nranks()
    if world < 2:
        pytest.skip("one visible GPU: the RCCL transport between devices cannot run here")
    n, grid = CASES[world]
    r = _launch_nccl(world, [os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(degree), "3",
                             ",".join(map(str, n)), ",".join(map(str, grid))] +
                     ([{True: "source", "asym": "asym"}[source]] if source else []))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import run_case
    _, us, ss = run_case(n, degree, 3, None, source)
    devices = set()
    for rank in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % rank))
        assert int(d["staged"]) == 0 and int(d["bytes_sent"]) > 0      # device buffers handed to RCCL
        devices.add(int(d["device"]))
        start, bn = d["start"], d["n"]
        idx = []
        for kz in range(start[2], start[2] + bn[2]):
            for j in range(start[1], start[1] + bn[1]):
                for i in range(start[0], start[0] + bn[0]):
                    cube = i + n[0] * (j + n[1] * kz)
                    idx.extend(cube * 6 + k for k in range(6))
        idx = np.array(idx)
        assert np.isfinite(d["u"]).all() and np.abs(d["u"]).max() > 0
        assert np.array_equal(d["u"], us[idx]), "velocity differs from the single-block run (rank %d)" % rank
        assert np.array_equal(d["s"], ss[idx]), "stress differs from the single-block run (rank %d)" % rank
    assert len(devices) == world, "one rank per GPU"


def test_bench_over_rccl(gpu):
    """bench.py as the driver launches it for N > 1, on N real devices: one JSON line, n_gpus = N, RCCL transport,
    the closed-form payload, the per-rank step times, the grid-size sweep."""
    world = _nranks()
    if world < 2:
        pytest.skip("one visible GPU")
    r = _launch_nccl(world, [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
                             "--cubes", "16", "--grid-sweep", "480,512", "--c4-cubes", "32", "--c4-steps", "3"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    steps = out["steps"]
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["cells"] == world * 16 ** 3 * 6
    grid = out["config"]["block_grid"]
    assert int(np.prod(grid)) == world
    h = out["halo"]
    assert h["transport"] == "nccl"
    assert h["exchanges_per_step"] == (6 * steps + 1) / steps
    # every block has one neighbour per cut axis; a side of a 16^3 block carries 16*16*2 facets * 15 nodes * 3 values
    # (velocity, or T.n of a stress) * 8 B in each of the 6 exchanges of a step (+ the one a run starts with)
    face = 16 * 16 * 2 * 15 * 3 * 8
    nsides = sum(1 for g in grid if g > 1)
    assert all(v == nsides * face * (6 * steps + 1) / steps for v in h["bytes_sent_per_step"]), h["bytes_sent_per_step"]
    assert len(out["rank_ms_per_step"]["per_rank"]) == world
    sw = h["grid_blocks_sweep_ms_per_step"]
    assert sorted(sw) == ["480", "512"] and all(v > 0 for v in sw.values())


def test_config3_golden_split_2x2x2_on_one_device(gpu):
    """Config 4's partition at production block widths, oracle-backed: config 3's input (64^3 cubes x 6 tets, P4, the
    eigenmode) cut into the 2 x 2 x 2 grid of eight 32^3 blocks on ONE device - group-thick x shells, item lists of the
    FIRST / SECOND regions, `T.n` ghost records on three sides of every block - stepped through the pipelined schedule
    with device copies as transport, against the oracle's golden of the unsplit mesh (fullsize_c3.npz: sampled cells
    and slab sums of every field)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    import bench
    from seigen_amd import BoxMesh, _lib
    from seigen_amd.backend import HipBlock
    from seigen_amd.mesh import Partition
    from tests import fullsize_cases as fc
    from tests.test_harness_gpu import _LocalExchange
    c = fc.C3
    N, P = c["n"], c["P"]
    gold = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_c3.npz"))
    grid, world = (2, 2, 2), 8
    parts = [Partition((N, N, N), r, world, grid) for r in range(world)]
    blocks = []

    class Shim(object):      # what bench.fill_initial_condition needs of a solver object
        pass

    for p in parts:
        mesh = BoxMesh(N, N, N, 1.0, 1.0, 1.0)
        mesh.set_partition(p)
        b = HipBlock(3, P, p.n, mesh.h, [p.start[a] * mesh.h[a] for a in range(3)], "left", p.nbr_mask)
        b.set_params(c["rho"], c["dt"], c["lam"], c["mu"])
        sh = Shim()
        sh.block, sh.mesh, sh.degree = b, mesh, P
        bench.fill_initial_condition(sh, c["dt"])
        blocks.append(b)
    ex = _LocalExchange(blocks, parts)
    ex.step(c["steps"], True)

    def cells_of(p):
        ax = [np.arange(p.start[a], p.start[a] + p.n[a]) for a in range(3)]
        cube = (ax[0][None, None, :] + N * (ax[1][None, :, None] + N * ax[2][:, None, None])).reshape(-1)
        return (cube[:, None] * 6 + np.arange(6)[None, :]).reshape(-1)

    sel = [cells_of(p) for p in parts]
    tol = dict(u=1e-10, s=1e-10, sh=1e-9, uh=1e-7)
    for name, f in (("u", _lib.FIELD_U), ("s", _lib.FIELD_S), ("uh", _lib.FIELD_UH), ("sh", _lib.FIELD_SH)):
        shape = blocks[0].field_shape(f)[1:]
        full = np.empty((6 * N ** 3,) + tuple(shape))
        for b, s in zip(blocks, sel):
            full[s] = b.get_field(f)
        want, want_layers = gold[name], gold[name + "_layers"]
        scale = np.abs(want).max()
        lscale = np.abs(want_layers).max()
        if name == "uh":      # the UH buffer holds w = dt u1 + dt^3/24 utemp (csrc/stages.cpp; tests/test_fullsize_oracle_gpu.py _compare)
            dt, c3 = c["dt"], c["dt"] ** 3 / 24.0
            scale = lscale = (dt * tol["u"] * np.abs(gold["u"]).max() + c3 * tol["uh"] * np.abs(gold["uh"]).max()) / tol["uh"]
            want, want_layers = dt * gold["u"] + c3 * gold["uh"], dt * gold["u_layers"] + c3 * gold["uh_layers"]
        assert np.isfinite(full).all() and scale > 0
        err = np.abs(full[gold["cells"]] - want).max() / scale
        lay = fc.layer_sums(full, N)
        lerr = np.abs(lay - want_layers).max() / max(lscale, scale)
        assert err < tol[name] and lerr < 30 * tol[name], (name, err, lerr)
        del full
    for b in blocks:
        b.close()
