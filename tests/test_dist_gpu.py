"""The multi-process path on hardware: two ranks (torch.distributed.run) sharing ONE MI355X, gloo
process group with host-staged halos (seigen_amd/parallel.py), through the public solver class.
Everything of the N > 1 path except the RCCL transport itself runs here: process-group set-up,
partition, exchanger, interior/boundary launches, bench.py's timing and reduction.  The result
must equal the single-block run bitwise."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(nproc, script_args, timeout=240):
    env = dict(os.environ, SEIGEN_DIST_BACKEND="gloo", SEIGEN_HIP_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="2", SEIGEN_TEST_HANG_DUMP="200")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("degree,n,grid,world,source", [
    (4, (16, 4, 4), (1, 1, 2), 2, False),      # MFMA path, z split
    (3, (16, 4, 2), (1, 2, 1), 2, False),      # MFMA path, y split
    (2, (4, 4, 4), (2, 1, 1), 2, False),       # generic path, x split
    (4, (16, 4, 4), (1, 2, 2), 4, False),      # four ranks on one device
    (4, (16, 4, 4), (1, 2, 2), 4, True),       # config 4's shape: 3-D source + sponge across blocks
    (2, (6, 4, 4), (2, 1, 2), 4, True),
    (4, (16, 4, 4), (1, 1, 2), 2, "asym"),     # only rank 0 is handed a non-symmetric source: both must leave symmetric storage
    (3, (16, 2, 4), (1, 1, 4), 4, "asym"),
])
def test_two_processes_one_gpu_bitwise(gpu, tmp_path, degree, n, grid, world, source):
    r = _launch(world, [os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), str(degree), "3",
                        ",".join(map(str, n)), ",".join(map(str, grid))] +
                ([{True: "source", "asym": "asym"}[source]] if source else []))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import run_case
    _, us, ss = run_case(n, degree, 3, None, source)
    single = {"u": us, "s": ss}
    for rank in range(world):
        d = np.load(tmp_path / ("rank%d.npz" % rank))
        assert int(d["staged"]) == 1 and int(d["bytes_sent"]) > 0
        start, bn = d["start"], d["n"]
        idx = []
        for kz in range(start[2], start[2] + bn[2]):
            for j in range(start[1], start[1] + bn[1]):
                for i in range(start[0], start[0] + bn[0]):
                    cube = i + n[0] * (j + n[1] * kz)
                    idx.extend(cube * 6 + k for k in range(6))
        idx = np.array(idx)
        assert np.isfinite(d["u"]).all() and np.abs(d["u"]).max() > 0
        assert np.array_equal(d["u"], single["u"][idx]), "velocity differs from the single-block run (rank %d)" % rank
        assert np.array_equal(d["s"], single["s"][idx]), "stress differs from the single-block run (rank %d)" % rank


def test_bench_two_ranks_one_gpu(gpu):
    """bench.py as the driver launches it for N > 1 (here: 2 ranks on one device, gloo)."""
    r = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cubes", "16"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["cells"] == 2 * 16 ** 3 * 6
    assert "cpu_baseline" not in out
    _check_halo_block(out, 2)
    # the persistent-grid sweep of the launches that overlap an exchange (tuning data for the first real multi-GPU run)
    sw = out["halo"]["grid_blocks_sweep_ms_per_step"]
    assert sorted(sw) == ["448", "480", "496", "512"] and all(v > 0 for v in sw.values())


def _check_halo_block(out, world):
    h = out["halo"]
    steps = out["steps"]
    # six exchanges per step + the one of the first stage's input that every _advance() call starts with
    assert h["transport"] == "host-staged/gloo" and h["exchanges_per_step"] == (6 * steps + 1) / steps
    for key in ("pack_ms_per_step", "bytes_sent_per_step", "exposed_wait_ms_per_step", "kernel_ms_per_step"):
        assert len(h[key]) == world and all(v >= 0 for v in h[key]), (key, h[key])
    # 1x1x2 grid of 16^3 blocks, P4: one side of 16*16*2 facets * 15 nodes * 3 comps (velocity, or T.n of a stress)
    face = 16 * 16 * 2 * 15 * 3 * 8
    assert all(v == face * (6 * steps + 1) / steps for v in h["bytes_sent_per_step"]), h["bytes_sent_per_step"]
    assert all(v > 0 for v in h["pack_ms_per_step"])
    # per-kernel accounting counts every stage once per step although a split stage is two launches
    r = out["roofline"]
    assert all(v["launches"] % out["steps"] == 0 for v in r["kernels"].values())
    assert 0 < r["frac_physical"] < r["frac"] < 1


def test_bench_starts_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no torchrun environment (how the driver starts N = 1) must start
    the two ranks itself and relay rank 0's line; a failing rank must fail the command."""
    env = dict(os.environ, SEIGEN_DIST_BACKEND="gloo", SEIGEN_HIP_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # "--n" (the short spelling of --cubes) must survive the launcher, whose own parser would call it ambiguous
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "16"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["block_grid"] == [1, 1, 2]
    _check_halo_block(out, 2)
    # a rank that cannot run (degree 9 does not exist) makes the whole command fail
    r = subprocess.run(cmd + ["--degree", "9"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rccl_transport_on_one_rank(gpu):
    """The "nccl" branch of HaloExchanger (RCCL send/receive of device buffers on the launch stream) on the
    test box's single GPU: one rank that is its own z-neighbour (tests/rccl_self_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("SEIGEN_DIST_BACKEND", "SEIGEN_HIP_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_self_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("rccl self-exchange ok") == 2, r.stdout


def test_native_rccl_exchange_on_one_rank(gpu):
    """The exchange inside the library (csrc/comm.cpp: sg_comm_init, then sg_step runs stages, packs and grouped
    ncclSend / ncclRecv for all steps in one call) on the test box's single GPU, bitwise against the host-driven
    exchanger (tests/rccl_native_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("SEIGEN_DIST_BACKEND", "SEIGEN_HIP_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_native_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("native rccl exchange ok") == 2, r.stdout


def test_bench_one_rank_under_torchrun_with_rccl(gpu):
    """bench.py as the driver launches it for N > 1, with the RCCL process group really initialised
    (a world of one rank: collectives and the reductions of the line run over RCCL)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("SEIGEN_DIST_BACKEND", "SEIGEN_HIP_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "2", "--warmup", "1", "--cubes", "16", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["value"] > 0



def test_bench_config4_workload_on_ranks_sharing_one_gpu(gpu):
    """`bench.py --workload c4` (BASELINE config 4: explosive source, fixed global size, block-split) with a small
    global mesh on 4 ranks that share the test box's GPU (gloo, host-staged): the line names the workload, reports
    strong scaling, per-rank step times and the halo block; the source is found through the support-box scan."""
    r = _launch(4, [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "c4", "--c4-cubes", "32", "--steps", "3",
                    "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["scaling"] == "strong" and out["value"] > 0
    assert "config 4" in out["config"]["workload"] and "32^3 cubes" in out["config"]["workload"]
    assert out["config"]["cells"] == 6 * 32 ** 3 and int(np.prod(out["config"]["block_grid"])) == 4
    assert len(out["rank_ms_per_step"]["per_rank"]) == 4
    assert 0 < out["rank_ms_per_step"]["min"] <= out["rank_ms_per_step"]["max"]
    assert len(out["halo"]["kernel_ms_per_step"]) == 4 and "host_blocked_ms_per_step" in out["halo"]
    # every stage counted once per step although split stages are two concurrent launches
    assert sum(out["roofline"]["stage_avg_ms"]) <= out["ms_per_step"] * 1.5


def test_bench_line_carries_the_other_configs(gpu):
    """One GPU: the bench line's "configs" object - BASELINE configs measured in the same job as the
    headline, each with value, ms_per_step, steps and the algorithmic and physical roofline fractions, the kernels named by
    the library; here configs 1 and 5 behind a small headline (the default run adds c2, config 4's share and the
    reference's 2-D N = 256 protocol), without the CPU baselines."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "8", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--configs", "c1,c5"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and "cpu_baseline" not in out
    assert out["roofline"]["kernel"].startswith("sg::mfma_stage_") and out["roofline"]["kernel"] in out["roofline"]["kernels"]
    assert sorted(out["configs"]) == ["c1", "c5"]
    for key, cells in (("c1", 3200), ("c5", 92686)):
        c = out["configs"][key]
        assert "error" not in c, c
        assert c["cells"] == cells and c["value"] > 0 and c["ms_per_step"] > 0 and c["steps"] > 0
        rf = c["roofline"]
        assert 0 < rf["frac_physical"] < rf["frac"] < 1 and rf["bound"] == "hbm"
        # five kernels: F plain (UH1), F fused (U1), F fused without the self term (UTEMP), G plain (STEMP, SH1), G fused (S1)
        assert rf["dominant_kernel"].startswith("sg::tile2d_stage<") and len(rf["kernels"]) == 5
        assert "cpu_baseline" not in c


def test_bench_line_config4_share_variants_and_degree_comparison(gpu):
    """The entries VERDICT r05 item 3 asked for, at a test's size: config 4's share from a smooth NON-ZERO state, the same
    block (re-used, not re-allocated) with the reference-style sponge strips on five faces, and one row of the reference's
    spatial-degree comparison (2-D N = 256, README.md:22-31)."""
    env = dict(os.environ, SEIGEN_BENCH_C4_SHARE_CUBES="32")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "8", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--configs", "c4_share,c4_share_sponge,ref_strong_2d_N256_P1_T2"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert sorted(out["configs"]) == ["c4_share", "c4_share_sponge", "ref_strong_2d_N256_P1_T2"]
    plain, sponge, p1 = (out["configs"][k] for k in ("c4_share", "c4_share_sponge", "ref_strong_2d_N256_P1_T2"))
    for c in (plain, sponge):
        assert "error" not in c, c
        assert c["cells"] == 6 * 32 ** 3 and c["value"] > 0 and "smooth non-zero" in c["workload"]
        assert 0 < c["roofline"]["frac_physical"] < c["roofline"]["frac"] < 1
        assert c["roofline"]["dominant_kernel"].startswith("sg::mfma_stage_")
    assert "sigma = 1000" in sponge["workload"] and "sigma" not in plain["workload"]
    assert sponge["ms_per_step"] > plain["ms_per_step"]          # the sponge costs something: it was really applied
    assert "error" not in p1, p1
    assert p1["degree"] == 1 and p1["steps"] == 1024 and p1["cells"] == 2 * 256 * 256 and "22-31" in p1["workload"]
    assert p1["u_error"] < 1e-2 and 0 < p1["roofline"]["frac_physical"] < p1["roofline"]["frac"] < 1
