"""The native in-library halo exchange (csrc/comm.cpp) with 2, 4 and 8 REAL ranks - on the test box's one GPU, through a
test-double transport.

Real RCCL refuses two ranks on one device, and no multi-GPU node is available to the tests, so until now comm.cpp had
only ever run with a single rank that is its own neighbour.  Here every rank is a thread of control of its own (fresh
worker processes; the 8-rank case runs two ranks per process to stay inside the box's limit of GPU processes) that drives
the C-ABI directly - sg_comm_check, sg_comm_init, sg_comm_selftest, one sg_step(n) - while `SEIGEN_RCCL_LIB` binds the
library's nine RCCL entry points to tests/fake_rccl (shared memory + host staging, RCCL's pairing and ordering rules;
checked on its own in tests/test_fake_rccl.py).  What runs for the first time between DIFFERENT ranks: ncclCommInitRank
across processes, peers on all three axes, the facing-side order of the receives, the statistics - and the result must
equal the single-block run BITWISE (the reference's exchange: seigen/elastic.py:404-436, ParLoopHaloEnd in
tests/tiling/utils.py:143-144).  The double has two modes: blocking (every call completes its transfers before it
returns: any number of ranks per process) and ASYNC (the transfers are work on the library's stream - copies and host
functions - and the calls return at once, as RCCL's do: the two-stream schedule of a split stage then runs concurrently
with the exchange instead of being serialised by the transport).  The second half runs the host-side agreement of seigen_amd/parallel.py::NativeExchanger
through the solver class (gloo process group, SEIGEN_HALO_NATIVE=force): a failure injected into ONE rank - argument
check, communicator, self-test - must put EVERY rank on the host-driven exchanger, within a timeout, with the same
bitwise result."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def fake():
    from fake_rccl.build import build
    return build()


def _env(fake, tmp_path, **extra):
    env = dict(os.environ, SEIGEN_RCCL_LIB=fake, FAKE_RCCL_TIMEOUT_S="60", FAKE_RCCL_LOG=str(tmp_path / "fake"),
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    env.pop("FAKE_RCCL_HOST", None)
    env.update(extra)
    return env


def _spawn_workers(fake, tmp_path, world, layout, grid, n, degree, steps, dtype, scenario, **extra_env):
    procs, first = [], 0
    for k in layout:
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tests", "native_exchange_worker.py"), str(tmp_path), str(world), str(first),
             str(k), ",".join(map(str, grid)), ",".join(map(str, n)), str(degree), str(steps), dtype, scenario],
            cwd=ROOT, env=_env(fake, tmp_path, **extra_env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        first += k
    assert first == world
    errs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            p.kill()
            so, se = p.communicate()
            se += "\n[killed after the timeout]"
        if p.returncode != 0:
            errs.append(so[-1500:] + se[-3000:])
    assert not errs, "\n-----\n".join(errs)


def _single_block(n, degree, steps, dtype, scenario):
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    from native_exchange_worker import setup_block
    blk = HipBlock(3, degree, n, [1.0 / n[a] for a in range(3)], [0.0] * 3, "left", 0, dtype=dtype)
    setup_block(blk, n, degree, scenario)
    blk.step(steps)
    blk.sync()
    out = {"u": blk.get_field(_lib.FIELD_U), "s": blk.get_field(_lib.FIELD_S), "uh": blk.get_field(_lib.FIELD_UH)}
    blk.close()
    return out


def _block_cells(n, start, bn):
    idx = []
    for kz in range(start[2], start[2] + bn[2]):
        for j in range(start[1], start[1] + bn[1]):
            for i in range(start[0], start[0] + bn[0]):
                cube = i + n[0] * (j + n[1] * kz)
                idx.extend(cube * 6 + k for k in range(6))
    return np.array(idx)


@pytest.mark.parametrize("world,layout,grid,n,degree,dtype,scenario,mode", [
    (2, [1, 1], (1, 1, 2), (16, 4, 4), 4, "f64", "plain", "blocking"),         # slab, the headline element
    (2, [1, 1], (2, 1, 1), (32, 2, 2), 4, "f64", "source", "blocking"),        # x split: shells of whole layout groups
    (4, [1, 1, 1, 1], (2, 2, 1), (32, 4, 2), 4, "f64", "source", "blocking"),  # 2 x 2 x 1, P4, source + sponge across the blocks
    (4, [1, 1, 1, 1], (1, 2, 2), (16, 4, 4), 2, "f64", "source", "blocking"),  # P2
    (4, [2, 2], (1, 1, 4), (16, 2, 8), 3, "f32", "plain", "blocking"),         # inner blocks with two neighbours on one axis, FP32
    (8, [2, 2, 2, 2], (2, 2, 2), (32, 4, 4), 4, "f64", "source", "blocking"),  # config 4's grid: 2 x 2 x 2, peers on three axes
    (8, [2, 2, 2, 2], (2, 2, 2), (8, 4, 4), 2, "f64", "source", "blocking"),   # P2, generic-kernel sized blocks
    # the double's ASYNC mode (one rank per process): the transfers are stream-ordered work and the calls return at once, as
    # RCCL's do - the SECOND launches on the second stream and the next stage's FIRST really run beside / behind them
    (2, [1, 1], (1, 1, 2), (16, 4, 4), 4, "f64", "source", "async"),
    (4, [1, 1, 1, 1], (2, 2, 1), (32, 4, 2), 4, "f64", "source", "async"),
    (4, [1, 1, 1, 1], (1, 2, 2), (32, 8, 8), 3, "f64", "plain", "async"),       # larger blocks: launches long enough to overlap
    (4, [1, 1, 1, 1], (1, 1, 4), (16, 2, 8), 2, "f32", "source", "async"),
])
def test_native_exchange_between_ranks_bitwise(gpu, fake, tmp_path, world, layout, grid, n, degree, dtype, scenario, mode):
    steps = 3
    extra = {"FAKE_RCCL_ASYNC": "1", "FAKE_RCCL_SLOT_BYTES": "1048576"} if mode == "async" else {}
    _spawn_workers(fake, tmp_path, world, layout, grid, n, degree, steps, dtype, scenario, **extra)
    single = _single_block(n, degree, steps, dtype, scenario)
    assert np.isfinite(single["u"]).all() and np.abs(single["u"]).max() > 0
    for rank in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % rank))
        assert int(d["selftest"]) == 0 and int(d["nsides"]) >= 1 and int(d["version"]) // 10000 == 2
        idx = _block_cells(n, d["start"], d["n"])
        for key in ("u", "s", "uh"):
            assert np.array_equal(d[key], single[key][idx]), "%s differs from the single-block run (rank %d)" % (key, rank)
        # the double really carried it: one message per side and exchange (+ the two self-test exchanges)
        log = json.loads(open(str(tmp_path / "fake") + ".rank%d" % rank).read().splitlines()[-1])
        nex = int(d["exchanges"]) + 2
        assert log["host_mode"] == 0 and log["sends"] == log["recvs"] == nex * int(d["nsides"]) and log["groups"] == nex
        assert log["async"] == (1 if mode == "async" else 0)
        assert log["bytes_sent"] >= int(d["bytes_sent"]) > 0


def test_config3_golden_through_the_native_exchange_on_eight_ranks(gpu, fake, tmp_path):
    """Config 4's partition at production block widths, oracle-backed, through the exchange INSIDE the library: BASELINE
    config 3's input (64^3 cubes x 6 tets, P4, the eigenmode) on the 2 x 2 x 2 grid of eight 32^3 blocks - eight ranks (two
    per worker process), each with three neighbours, group-thick x shells, `T.n` ghost records on three sides - stepped by
    one sg_step per rank over the transport double, against the ORACLE's golden of the unsplit mesh (tests/golden/
    fullsize_c3.npz: sampled cells and slab sums of every field).  tests/test_multigpu_gpu.py checks the same partition
    with device copies between blocks of one process; this is the path the first 8-GPU job takes."""
    from tests import fullsize_cases as fc
    c = fc.C3
    N, world = c["n"], 8
    _spawn_workers(fake, tmp_path, world, [2, 2, 2, 2], (2, 2, 2), (N, N, N), c["P"], c["steps"], "f64", "c3golden")
    gold = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_c3.npz"))
    ranks = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    assert all(int(d["selftest"]) == 0 and int(d["nsides"]) == 3 for d in ranks)
    tol = dict(u=1e-10, s=1e-10, sh=1e-9, uh=1e-7)
    seen = np.zeros(len(gold["cells"]), dtype=int)
    for d in ranks:
        seen[d["g_which"]] += 1
    assert (seen == 1).all(), "every sampled cell of the golden lies in exactly one block"
    for name in ("u", "s", "uh", "sh"):
        want, want_layers = gold[name], gold[name + "_layers"]
        scale, lscale = np.abs(want).max(), np.abs(want_layers).max()
        if name == "uh":      # the UH buffer holds w = dt u1 + dt^3/24 utemp (csrc/stages.cpp; tests/test_fullsize_oracle_gpu.py _compare)
            dt, c3 = c["dt"], c["dt"] ** 3 / 24.0
            scale = lscale = (dt * tol["u"] * np.abs(gold["u"]).max() + c3 * tol["uh"] * np.abs(gold["uh"]).max()) / tol["uh"]
            want, want_layers = dt * gold["u"] + c3 * gold["uh"], dt * gold["u_layers"] + c3 * gold["uh_layers"]
        got = np.empty_like(want)
        layers = np.zeros_like(want_layers)
        for d in ranks:
            got[d["g_which"]] = d["g_" + name]
            z0, nz = int(d["start"][2]), int(d["n"][2])
            layers[z0:z0 + nz] += d["g_" + name + "_layers"].reshape(nz, -1)
        assert np.isfinite(got).all() and scale > 0
        err = np.abs(got - want).max() / scale
        lerr = np.abs(layers - want_layers).max() / max(lscale, scale)
        assert err < tol[name] and lerr < 30 * tol[name], (name, err, lerr)


def test_two_faces_between_one_pair_of_ranks_pair_by_facing_side(gpu, fake, tmp_path):
    """Two blocks around a wrapped z axis: each rank's z- and z+ sides both lead to the other rank.  RCCL pairs the two
    messages of such a pair in posting order; side s must get what the peer sent from its side s ^ 1 (comm.cpp posts the
    receives in the order of the facing sides) - the mirror image would arrive otherwise, silently."""
    _spawn_workers(fake, tmp_path, 2, [1, 1], (1, 1, 2), (16, 4, 4), 3, 2, "f64", "wrap")
    d = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(2)]
    for r in range(2):
        assert int(d[r]["selftest"]) == 0 and np.isfinite(d[r]["u"]).all()
        for kind in (0, 1):
            for s in (4, 5):
                got, want = d[r]["recv_%d_%d" % (kind, s)], d[1 - r]["send_%d_%d" % (kind, s ^ 1)]
                assert want.any() and np.array_equal(got, want), "rank %d side %d did not receive the peer's facing side" % (r, s)
            assert not np.array_equal(d[r]["recv_%d_4" % kind], d[r]["recv_%d_5" % kind])


# ---- the host-side agreement (seigen_amd/parallel.py::NativeExchanger) through the solver class ---------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("fault,native", [
    (None, True),                 # no fault: every rank runs the exchange inside the library
    ("check", False),             # rank 1's peers do not match its neighbour mask: refused locally (sg_comm_check)
    ("init", False),              # rank 2's ncclCommInitRank fails alone
    ("selftest", False),          # rank 3 receives corrupted traces: its self-test counts mismatches
])
def test_ranks_agree_on_the_exchanger(gpu, fake, tmp_path, fault, native):
    world, grid, n, degree = 4, (1, 2, 2), (16, 4, 4), 4
    extra = {"SEIGEN_DIST_BACKEND": "gloo", "SEIGEN_HIP_DEVICE": "0", "SEIGEN_HALO_NATIVE": "force", "SEIGEN_TEST_HANG_DUMP": "200",
             "FAKE_RCCL_TIMEOUT_S": "30"}
    if fault == "init":
        extra["FAKE_RCCL_FAIL_INIT"] = "2"
    if fault == "selftest":
        extra["FAKE_RCCL_CORRUPT_RECV"] = "3"
    if fault == "check":
        extra["SEIGEN_TEST_BAD_PEERS_RANK"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path),
           str(degree), "3", ",".join(map(str, n)), ",".join(map(str, grid)), "source"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(fake, tmp_path, **extra), capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    from dist_worker import run_case
    _, us, ss = run_case(n, degree, 3, None, True)
    for rank in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % rank))
        # every rank ended up on the SAME exchanger: the native one, or - after a failure on one rank - the host-driven one
        assert int(d["native"]) == int(native), "rank %d: native = %d" % (rank, int(d["native"]))
        assert int(d["staged"]) == (0 if native else 1) and int(d["bytes_sent"]) > 0
        idx = _block_cells(n, d["start"], d["n"])
        assert np.array_equal(d["u"], us[idx]) and np.array_equal(d["s"], ss[idx]), "rank %d differs from the single block" % rank
        if native:
            assert os.path.basename(str(d["library"])) == "libfake_rccl.so"


@pytest.mark.parametrize("args,cells,scaling", [
    (["--gpus", "4", "--steps", "3", "--warmup", "1", "--cubes", "16"], 4 * 6 * 16 ** 3, "weak"),
    (["--gpus", "4", "--workload", "c4", "--c4-cubes", "32", "--steps", "3", "--warmup", "1"], 6 * 32 ** 3, "strong"),
])
def test_bench_reports_the_native_exchange(gpu, fake, tmp_path, args, cells, scaling):
    """bench.py as the driver launches it for N > 1, with the exchange INSIDE the library (the default of every RCCL run)
    carried by the transport double in its async mode: the line's halo block - driver, transport, which RCCL, payload per
    step against the closed form, waits, the grid sweep of the weak-scaling workload - has only ever been filled by the
    host-driven exchanger before (tests/test_dist_gpu.py); the first 8-GPU job goes through exactly this code."""
    extra = {"SEIGEN_DIST_BACKEND": "gloo", "SEIGEN_HIP_DEVICE": "0", "SEIGEN_HALO_NATIVE": "force", "FAKE_RCCL_ASYNC": "1",
             "FAKE_RCCL_SLOT_BYTES": "1048576", "FAKE_RCCL_TIMEOUT_S": "60"}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=_env(fake, tmp_path, **extra), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["scaling"] == scaling and out["value"] > 0 and out["config"]["cells"] == cells
    h = out["halo"]
    assert h["driver"] == "native" and h["transport"] == "nccl"
    assert os.path.basename(h["rccl_library"]) == "libfake_rccl.so" and h["rccl_version"] // 10000 == 2
    steps = out["steps"]
    assert h["exchanges_per_step"] == (6 * steps + 1) / steps          # + the exchange of the first stage's input per sg_step call
    for key in ("pack_ms_per_step", "bytes_sent_per_step", "exposed_wait_ms_per_step", "kernel_ms_per_step"):
        assert len(h[key]) == 4 and all(v >= 0 for v in h[key]), (key, h[key])
    assert all(v > 0 for v in h["bytes_sent_per_step"]) and all(v == 0 for v in h["host_blocked_ms_per_step"])
    if scaling == "weak":
        # 1 x 2 x 2 grid of 16^3 blocks, P4: two sides of 16 * 16 * 2 facets * 15 nodes * 3 components each
        face = 16 * 16 * 2 * 15 * 3 * 8
        assert out["config"]["block_grid"] == [1, 2, 2]
        assert all(v == 2 * face * (6 * steps + 1) / steps for v in h["bytes_sent_per_step"]), h["bytes_sent_per_step"]
        sw = h["grid_blocks_sweep_ms_per_step"]
        assert sorted(sw) == ["448", "480", "496", "512"] and all(v > 0 for v in sw.values())
    assert len(out["rank_ms_per_step"]["per_rank"]) == 4
