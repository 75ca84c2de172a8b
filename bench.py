#!/usr/bin/env python
"""Headline benchmark: M DoF-updates/s of the explicit velocity-stress LF4 step,
3-D elastic eigenmode, P=4, on a structured tetrahedral mesh (BASELINE.json
config 3: 64^3 cubes x 6 tets, P4, FP64), one mesh block per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT torchrun and with --gpus N > 1, this process starts the N ranks itself (a child
`python -m torch.distributed.run ...`, before anything here touches a GPU), relays rank 0's JSON
line and exits non-zero if any rank failed - the reference's protocol is
`mpiexec -n NP python eigenmode_bench.py ...` (tests/eigenmode/README.md:7-13).

One "step" = one full LF4 timestep (six fused HIP launches = the reference's
eight solves + two assigns, seigen/elastic.py:291-304) over the whole mesh.
DoF-updates = (U dofs + S dofs) * steps  (SURVEY.md 8d).  Weak scaling: every
GPU owns an n^3-cube block.  Prints ONE JSON line on rank 0.

--workload c4 measures BASELINE config 4 instead (3-D explosive source, 256^3 cubes in total, P4, FP64, zero initial
state, box-Ricker stress source, block-split over the ranks: fixed global size, the protocol of
tests/eigenmode/README.md:7-13); it needs 4 or 8 GPUs (676 GB of fields; on one GPU it runs one rank's 128^3 share).
With 8 ranks and no --workload the weak-scaling line carries an extra object "config4" measured in the same job.

One GPU, default headline: the same JSON line carries "configs" - every other single-GPU configuration of BASELINE.json and
the reference's own benchmark protocol, measured after the headline in the same job (seigen_amd/harness/baseline_configs.py):
c1 (2-D eigenmode 40 x 40, P1), c2 (2-D explosive source 512^2, P2), c5 (Marmousi, P3), c4_share (one rank's 128^3 share of
config 4), ref_strong_2d_N256_P4_T2 (tests/eigenmode/README.md:7-13 through the solver class) - each with value, ms_per_step,
steps, the algorithmic and physical roofline fractions, the kernels as the library names them, and for c1 / c2 / c5 a CPU
baseline of the oracle's C port (oracle/baselines.py).  Each entry runs under its own deadline; none can lose
the headline.  The whole default run takes under a minute.

Watchdogs: every rank arms faulthandler.dump_traceback_later(--timeout): a rank that hangs in init or in an exchange
dumps its stacks and exits non-zero; the self-launching parent (which never touches a GPU) additionally kills the
child process group --timeout + 30 s after starting it.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

# The host driver of this pool only supports dmabuf IPC: without this RCCL (and torch's sharing of device tensors between
# processes) fails with `hipIpcGetMemHandle: invalid argument`.  It has to be in the environment before the HIP runtime
# starts - i.e. before torch or libseigen_hip are loaded - so it is set here, at import, if the launcher did not.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TFLOPS = 78.6        # MI355X datasheet FP64 matrix peak (256 CUs x 128 flop/clk x 2.4 GHz)
FP64_MFMA_SUSTAINED_TFLOPS = 52.4   # v_mfma_f64_16x16x4_f64 (the shape the kernels use) back to back, measured
                                    # (tools/ubench_fp64_mix.hip, profiles/r01).  Not a ceiling: a loop shaped like the
                                    # kernels' volume phase sustains 62 with VGPR accumulators, the 4x4x4_4b shape 74
                                    # alone (tools/ubench_mfma_insitu.hip, profiles/r02/mfma_small_tiles_experiment.txt)


def eigenmode3d_fields(X, t_u, t_s):
    """Analytic eigenmode of tests/eigenmode/eigenmode_3d.py:30-40 (rho=1, mu=0.25)."""
    mu, rho = 0.25, 1.0
    A = math.sqrt(2 * rho * mu)
    O = math.pi * math.sqrt(2 * mu / rho)
    pi = math.pi
    x, y, z = X[..., 0], X[..., 1], X[..., 2]
    cx, cy, cz = np.cos(pi * x), np.cos(pi * y), np.cos(pi * z)
    sx, sy, sz = np.sin(pi * x), np.sin(pi * y), np.sin(pi * z)
    c, s = math.cos(O * t_u), math.sin(O * t_s)
    u = np.stack([cx * (sy - sz) * c, cy * (sz - sx) * c, cz * (sx - sy) * c], axis=-1)
    T = np.zeros(X.shape[:-1] + (3, 3))
    T[..., 0, 0] = -A * sx * (sy - sz) * s
    T[..., 1, 1] = -A * sy * (sz - sx) * s
    T[..., 2, 2] = -A * sz * (sx - sy) * s
    return u, T


def fill_initial_condition(elastic, dt):
    """Chunked (one z-layer of cubes at a time) nodal interpolation + upload."""
    import ctypes as C
    from seigen_amd import _lib
    from seigen_amd.functionspace import block_config
    blk = elastic.block
    mesh = elastic.mesh
    part = mesh.partition
    nx, ny, nz = part.n
    cells_per_layer = nx * ny * 6
    lib = _lib.load()
    for k in range(nz):
        cfg = block_config(mesh, elastic.degree)
        cfg.n[2] = 1
        cfg.cube0[2] = part.start[2] + k
        X = np.empty((cells_per_layer, blk.nd, 3))
        _lib.check(lib.sg_block_node_coords(C.byref(cfg), elastic.degree, X.ctypes.data, X.nbytes))
        u, T = eigenmode3d_fields(X, 0.0, dt / 2.0)
        blk.set_field_range(_lib.FIELD_U, k * cells_per_layer, u)
        blk.set_field_range(_lib.FIELD_S, k * cells_per_layer, T)


def self_launch(ngpus, timeout_s):
    """--gpus N > 1 without a torchrun environment: start the ranks as a child process group and
    relay what they print.  Nothing in this parent has initialised HIP (no exec of a GPU process)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # the launcher's own parser abbreviates options even after the script name: "--n" would be "ambiguous" there
    argv = ["--cubes" if a == "--n" else ("--cubes=" + a[4:] if a.startswith("--n=") else a) for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    import signal
    # own session = own process group: the watchdog can take down the launcher and every rank with it
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s + 30.0)
    except subprocess.TimeoutExpired:
        print("bench.py: no result after %.0f s - killing the ranks (process group %d)" % (timeout_s + 30.0, proc.pid),
              file=sys.stderr)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10.0)
                break
            except subprocess.TimeoutExpired:
                continue
        sys.exit(124)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    for ln in out.splitlines():
        print(ln)
    sys.stdout.flush()
    if proc.returncode != 0:
        print("bench.py: a rank failed (torch.distributed.run exit code %d)" % proc.returncode, file=sys.stderr)
        sys.exit(proc.returncode)
    if len(lines) != 1:
        print("bench.py: expected one JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


def cpu_baseline(degree, budget_s=12.0):
    """The oracle's plain-C/OpenMP restatement of the reference path (oracle/c/seigen_oracle.c,
    kind "port") timed on this host's cores: 3-D eigenmode, N=16 (24 576 tets), same P, FP64.
    A bounded sample (~15 s of CPU work); baseline only.  (oracle/baselines.py holds the set-ups.)"""
    from oracle import baselines
    baselines._native_build()
    return baselines.config3(degree, budget_s)


def csrc_digest():
    """sha256 over the kernel / host sources the library is built from: identifies WHICH kernels a committed
    traffic measurement belongs to (tools/make_traffic_json.py stores the same digest)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "seigen_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "seigen_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(ROOT, "seigen_amd", "csrc", "*.hpp")) +
                   [os.path.join(ROOT, "seigen_amd", "csrc", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


class Comm(object):
    """The few collectives the bench needs (never on the data path)."""

    def __init__(self, dist, backend, world):
        self.dist, self.backend, self.world = dist, backend, world

    def _t(self, x):
        import torch
        return torch.tensor([float(x)], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")

    def barrier(self):
        if self.dist is not None:
            import torch
            torch.cuda.synchronize()
            self.dist.barrier()

    def reduce_max(self, x):
        if self.dist is None:
            return x
        t = self._t(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def gather(self, x):
        """one float per rank -> list over ranks (on every rank)"""
        if self.dist is None:
            return [float(x)]
        import torch
        t = self._t(x)
        outs = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return [o.item() for o in outs]


def build_config3(args, rank, world):
    """Weak scaling of BASELINE config 3: the unit-cube eigenmode on an (n gx, n gy, n gz) mesh of cell size 1/n."""
    from seigen_amd import ElasticLF4, BoxMesh
    from seigen_amd.mesh import Partition, _factor_grid
    n, P = args.n, args.degree
    grid = _factor_grid(world, 3, (n, n, n))
    gn = tuple(n * g for g in grid)
    mesh = BoxMesh(gn[0], gn[1], gn[2], gn[0] / n, gn[1] / n, gn[2] / n)
    mesh.set_partition(Partition(gn, rank, world, grid))
    elastic = ElasticLF4.create(mesh, "DG", P, dimension=3, solver="explicit", output=False, dtype=args.dtype)
    elastic.density, elastic.mu, elastic.l = 1.0, 0.25, 0.5
    elastic.dt = 0.5 * (1.0 / n) / 2 ** (P - 1)
    fill_initial_condition(elastic, elastic.dt)
    elastic.setup()
    elastic.block.set_source([], None)
    name = "3D eigenmode, %dx%dx%d cubes x 6 tets (%d^3 per GPU), DG P%d, %s, LF4" % (
        gn[0], gn[1], gn[2], n, P, "FP32" if args.dtype == "f32" else "FP64")
    return elastic, grid, gn, name, "weak"


def build_config4(args, rank, world, nsteps_source):
    """BASELINE config 4: 3-D explosive source, 256^3 cubes x 6 tets in total (fixed global size), P4, zero initial
    state, the explosive test's material (explosive_source_lf4.py:21-23) and Ricker wavelet (:37-38) in a
    4-cube box at the centre, blocks from Partition.  One rank: its 128^3 share alone (85 GB)."""
    from seigen_amd import ElasticLF4, BoxMesh, Expression, Function, Vp, cfl_dt
    from seigen_amd.mesh import Partition, _factor_grid
    P, h = args.degree, 2.5
    N = args.c4_cubes if world > 1 else args.c4_cubes // 2
    grid = _factor_grid(world, 3, (N, N, N))
    mesh = BoxMesh(N, N, N, N * h, N * h, N * h)
    mesh.set_partition(Partition((N, N, N), rank, world, grid))
    el = ElasticLF4.create(mesh, "DG", P, dimension=3, solver="explicit", output=False, dtype=args.dtype)
    el.density, el.mu, el.l = 1.0, 3600.0, 3599.3664
    el.dt = cfl_dt(h, Vp(el.mu, el.l, el.density), 0.05) / 2 ** (P - 1)
    c = 0.5 * N * h
    box = " && ".join("x[%d] >= %r && x[%d] <= %r" % (a, c - 2 * h, a, c + 2 * h) for a in range(3))
    # the wavelet of explosive_source_lf4.py:37-38, centred 40 steps into the run so that the timed steps carry it
    code = "%s ? (-1.0 + 2*a*pow(t - t0, 2))*exp(-a*pow(t - t0, 2)) : 0.0" % box
    z = "0.0"
    el.source_expression = Expression(((code, z, z), (z, code, z), (z, z, code)), a=159.42, t0=40 * el.dt, t=0)
    el.source_expression.support_box = ((c - 2 * h,) * 3, (c + 2 * h,) * 3)
    el.source_function = Function(el.S)          # zero; the per-step table below is what the kernels see
    el.setup()
    times = [el.dt * (k + 1) for k in range(nsteps_source)]
    el.upload_source(times)
    el._agree_on_stress_storage()
    name = "3D explosive source (config 4), %d^3 cubes x 6 tets in total, blocks %s, DG P%d, %s, LF4" % (
        N, "x".join(str(v) for v in mesh.partition.n), P, "FP32" if args.dtype == "f32" else "FP64")
    if world == 1:
        name += " - ONE rank's share of the 256^3 mesh (no neighbours)"
    return el, grid, (N, N, N), name, "strong"


def measure(elastic, args, comm, steps, warmup):
    """W untimed + K timed steps between barriers; returns per-stage device times, counters, timings."""
    import seigen_amd
    blk = elastic.block

    def sync():
        blk.sync()
        comm.barrier()

    sync()
    t0 = time.perf_counter()
    elastic._advance(warmup)
    sync()
    warm_s = comm.reduce_max(time.perf_counter() - t0)
    if steps is None:
        # long enough for clocks and power to settle (the package reaches its power cap within about
        # a second): at least 2 s of timed stepping, from the warm-up's own rate
        per_step = warm_s / max(warmup, 1)
        steps = int(min(max(math.ceil(2.2 / max(per_step, 1e-6)), 20), 20000))
        if args.n == 64 and args.degree == 4 and comm.world == 1:
            steps = max(steps, 250)
    ex = elastic._exchanger
    if ex is not None:
        ex.reset_stats(timing=True)
    blk.enable_timing(True)
    c0 = blk.counters()
    sync()
    t0 = time.perf_counter()
    elastic._advance(steps)
    blk.sync()
    t_own = time.perf_counter() - t0          # this rank alone (before the closing barrier)
    comm.barrier()
    t1 = time.perf_counter()
    c1 = blk.counters()
    blk.enable_timing(False)
    elapsed = comm.reduce_max(t1 - t0)
    probe = blk.get_field_range(seigen_amd._lib.FIELD_U, 0, 6)
    assert np.isfinite(probe).all(), "non-finite solution"
    return dict(steps=steps, elapsed=elapsed, own_ms_per_step=comm.gather(t_own / steps * 1e3), c0=c0, c1=c1,
                ranks=int(round(sum(comm.gather(1.0)))))


def stage_accounting(dim, sym):
    """Words per node and STAGE (UH1 STEMP U1 SH1 UTEMP S1): algorithmic (every input of the stage read once, every output
    written once, d x d stress) and what the kernels physically move when the stress is stored symmetric (d (d + 1) / 2
    of the d^2 lines).  The reference's eight solves and two assigns come to 8 words = 64 B (FP64) per
    DoF-update; this step needs fewer: g is linear, so stage UTEMP leaves w = dt u1 + dt^3/24 utemp (reading
    u1 beside sh1) and stage S1 is s0 + G(w) - no sh1, no second right-hand side (csrc/stages.cpp): 90 instead of 96
    words per node in 3-D = 60 B per DoF-update, 46 instead of 48 in 2-D = 61.3 B.  Rates and roofline fractions below are
    quoted on THESE bytes (what the launches have to move), the metric itself - DoF-updates per second - does not change."""
    d, sw = dim, dim * dim
    words = [sw + d, d + sw, sw + 3 * d, d + sw, sw + 2 * d, 2 * sw + d]
    sw6 = d * (d + 1) // 2
    phys = [sw6 + d, d + sw6, sw6 + 3 * d, d + sw6, sw6 + 2 * d, 2 * sw6 + d] if sym else list(words)
    return words, phys


def bytes_per_dof_update(dim, esz=8, sym=False):
    words, phys = stage_accounting(dim, sym)
    return sum(phys if sym else words) * float(esz) / (dim + dim * dim)


def kernel_table(blk, c0, c1, esz, region=0):
    """Per rocprofv3 kernel name - asked of the LIBRARY (sg_stage_kernel_name: the stage's own dispatch code names the
    instantiation it launches), not re-derived from switches - the device time of the timed region (hipEvent pairs
    around every launch), launches, algorithmic / physical bytes and rates.  Returns (kern, stage_ms, nst)."""
    from collections import OrderedDict
    nodes = blk.ncells * blk.nd
    ms = [c1["kernel_ms"][i] - c0["kernel_ms"][i] for i in range(6)]
    nl = [c1["launches"][i] - c0["launches"][i] for i in range(6)]
    nst = c1["steps"] - c0["steps"]
    words, words_phys = stage_accounting(blk.dim, blk.is_sym())
    groups = OrderedDict()
    for st in range(6):
        groups.setdefault(blk.stage_kernel_name(st, region), []).append(st)
    kern = OrderedDict()
    for name, stages in groups.items():
        tot_ms = sum(ms[i] for i in stages)
        # bytes per STEP, not per launch: a split stage (multi-GPU: FIRST + SECOND) covers the block once, and its
        # device time is counted once (the longer of the two concurrent launches)
        byts = sum(words[i] for i in stages) * nst * nodes * float(esz)
        byts_phys = sum(words_phys[i] for i in stages) * nst * nodes * float(esz)
        per = len(stages) * nst           # stage executions of this kernel in the timed region
        kern[name] = dict(ms=tot_ms, launches=sum(nl[i] for i in stages), avg_ms=tot_ms / max(per, 1), stages=stages,
                          gbs=byts / max(tot_ms, 1e-12) / 1e6, gbs_phys=byts_phys / max(tot_ms, 1e-12) / 1e6,
                          bytes_per_launch=byts / max(per, 1), bytes_phys_per_launch=byts_phys / max(per, 1))
    return kern, ms, nst


def halo_block(elastic, m, comm, backend):
    ex, c0, c1, nst = elastic._exchanger, m["c0"], m["c1"], m["steps"]
    st = ex.stats()
    ms = [c1["kernel_ms"][i] - c0["kernel_ms"][i] for i in range(6)]
    # the native exchange moves device buffers over its own RCCL communicator whatever the process group is
    return {"transport": "nccl" if getattr(ex, "native", False) else ("host-staged/%s" % backend if ex.staged else backend),
            # who drives the exchange: the library itself (csrc/comm.cpp: one C-ABI call per run of steps) or the
            # Python exchanger stage by stage (seigen_amd/parallel.py)
            "driver": "native" if getattr(ex, "native", False) else "python",
            # which RCCL the native exchange is bound to (sg_comm_library / sg_comm_version): the copy already in the
            # process, the system's, or the file SEIGEN_RCCL_LIB names
            "rccl_library": getattr(ex, "library", None), "rccl_version": getattr(ex, "version", None),
            "pack_ms_per_step": comm.gather((c1["halo_pack_ms"] - c0["halo_pack_ms"]) / nst),
            "bytes_sent_per_step": comm.gather((c1["halo_bytes_packed"] - c0["halo_bytes_packed"]) / nst),
            # everything a stage waited for its traces (same definition as the round-1/2 records) ...
            "exposed_wait_ms_per_step": comm.gather(st["exposed_wait_ms"] / nst),
            # ... and the transport's share of it (host-staged: without the wait for the rank's own launch and copies)
            "exposed_wait_transport_ms_per_step": comm.gather(st.get("exposed_wait_transport_ms", st["exposed_wait_ms"]) / nst),
            "host_blocked_ms_per_step": comm.gather(st.get("host_blocked_ms", 0.0) / nst),
            "exchanges_per_step": st["exchanges"] / nst,
            # split stages count once, with the longer of their two concurrent launches (sg_get_counters)
            "kernel_ms_per_step": comm.gather(sum(ms) / nst),
            "grid_blocks": os.environ.get("SEIGEN_HIP_GRID_BLOCKS", "default (15/16 of the block slots while exchanging)")}


C4_SHARE_CUBES = int(os.environ.get("SEIGEN_BENCH_C4_SHARE_CUBES", "128"))        # one rank's share of config 4's 256^3 cubes on 8 GPUs (tests: SEIGEN_BENCH_C4_SHARE_CUBES)
_C4_KEEP = {}


def secondary_config(key, with_cpu=True, threads=None, keep_c4=False):
    """One entry of the bench line's "configs" object: a BASELINE configuration other than the headline (or the
    reference's own benchmark protocol) through the solver class on this GPU - W untimed + K timed steps between
    synchronisations, per-kernel device times in a second pass - and, for c1 / c2 / c5, the oracle's C port on the host
    cores beside it (C1 in full, C2 20 steps, C5 50 steps)."""
    from seigen_amd.harness import baseline_configs as bc
    if key.startswith("ref_strong_2d_N256_P") and key.endswith("_T2"):
        # P4: the strong-scaling protocol (README.md:7-13); P1..P3: its spatial-degree comparison on the same mesh (:22-31)
        deg = int(key[len("ref_strong_2d_N256_P"):-len("_T2")])
        rec = bc.reference_strong_2d(256, deg, 2.0)
        t = rec["timestepping_s"] or rec["run_wall_s"]
        value = rec["dofs"] * rec["steps"] / t / 1e6
        bpu = bytes_per_dof_update(2)
        frac = value * 1e6 * bpu / 1e9 / HBM_PEAK_GBS
        rec.update({"workload": "the reference's %s on one device (tests/eigenmode/README.md:%s): 2D eigenmode "
                                "N=256, P%d, T=2.0, explicit; one warm-up run as pybench's warmups = 1, whole run(T) through the solver class"
                                % (("strong-scaling protocol", "7-13", deg) if deg == 4 else ("spatial-degree comparison", "22-31", deg)),
                    "value": value, "unit": "M DoF-updates/s", "ms_per_step": t / rec["steps"] * 1e3,
                    "roofline": {"bound": "hbm", "frac": frac, "frac_physical": frac * bytes_per_dof_update(2, sym=True) / bpu,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_dof_update": bpu,
                                 "note": "whole run: algorithmic bytes per DoF-update (bench.py stage_accounting) x value / peak; "
                                         "physical = with the stress stored symmetric"}})
        return rec
    steps, warm = {"c1": (20000, 1000), "c2": (2000, 50), "c5": (5000, 100), "c4_share": (20, 3), "c4_share_sponge": (20, 3)}[key]
    t0 = time.perf_counter()
    if key.startswith("c4_share"):
        # both entries on ONE block (85 GB, allocated once), from a smooth non-zero state: on the all-zero block rounds 1-5
        # measured the package clocks 5-8 % higher
        el, label = bc.config4_share(2 * steps + warm, n=C4_SHARE_CUBES, sponge=key.endswith("_sponge"), el=_C4_KEEP.pop("el", None))
    else:
        el, label = {"c1": bc.config1, "c2": bc.config2, "c5": bc.config5}[key](2 * steps + warm)
    blk = el.block
    setup_s = time.perf_counter() - t0
    el._advance(warm)
    blk.sync()
    t0 = time.perf_counter()
    el._advance(steps)
    blk.sync()
    elapsed = time.perf_counter() - t0
    dev_ms = blk.last_step_ms()
    dofs = blk.u_dofs + blk.s_dofs
    value = dofs * steps / elapsed / 1e6
    probe = blk.get_field_range(0, 0, 4)
    assert np.isfinite(probe).all(), "non-finite solution"
    # per-kernel device times: a second pass with an event pair around every launch (no graph replay there; the
    # headline figures above come from the untimed pass)
    words, words_phys = stage_accounting(blk.dim, blk.is_sym())
    blk.enable_timing(True)
    c0 = blk.counters()
    nt = max(min(steps, 200), 1)
    el._advance(nt)
    blk.sync()
    c1 = blk.counters()
    blk.enable_timing(False)
    kern, stage_ms, nst = kernel_table(blk, c0, c1, 8)
    dom = max(kern, key=lambda k: kern[k]["ms"])
    phys_ratio = sum(words_phys) / float(sum(words))
    bpu = bytes_per_dof_update(blk.dim)
    out = {"workload": label, "value": value, "unit": "M DoF-updates/s", "ms_per_step": elapsed / steps * 1e3,
           "device_ms_per_step": dev_ms / steps, "steps": steps, "warmup": warm, "dofs": int(dofs), "cells": int(blk.ncells),
           "dt": el.dt, "setup_s": setup_s,
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        # whole step: algorithmic bytes per DoF-update (stage_accounting: 60 B in 3-D, 61.3 B in 2-D) x value
                        "bytes_per_dof_update": bpu,
                        "achieved": value * 1e6 * bpu / 1e9, "frac": value * 1e6 * bpu / 1e9 / HBM_PEAK_GBS,
                        "frac_physical": value * 1e6 * bpu / 1e9 / HBM_PEAK_GBS * phys_ratio,
                        "dominant_kernel": dom, "dominant_kernel_frac": kern[dom]["gbs"] / HBM_PEAK_GBS,
                        "dominant_kernel_frac_physical": kern[dom]["gbs_phys"] / HBM_PEAK_GBS,
                        "kernels": {kk: {"avg_us": v["avg_ms"] * 1e3, "launches": v["launches"], "algorithmic_GBps": v["gbs"],
                                         "physical_GBps": v["gbs_phys"]} for kk, v in kern.items()},
                        "stage_avg_us_event_timed": [stage_ms[i] / max(nst, 1) * 1e3 for i in range(6)]}}
    if key == "c4_share" and keep_c4:
        _C4_KEEP["el"] = el          # the sponge variant follows: same block
    else:
        blk.close()
    del el, blk
    if with_cpu and key in ("c1", "c2", "c5"):
        from oracle import baselines
        out["cpu_baseline"] = {"c1": baselines.config1, "c2": baselines.config2, "c5": baselines.config5}[key](threads=threads)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: as many as make the timed region at least 2 s, e.g. 250 at config 3)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--cubes", type=int, default=64, help="cubes per axis per GPU (64 = BASELINE config 3)")
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("f64", "f32"), default="f64",
                    help="f64: the reference's precision (headline); f32: the separately reported FP32 second mode "
                         "(SURVEY 8d; 30 B per DoF-update on this step's 90 words per node)")
    ap.add_argument("--workload", choices=("c3", "c4"), default=None,
                    help="c3: weak scaling of config 3 (default); c4: BASELINE config 4 (256^3 in total, 4 or 8 GPUs; "
                         "one GPU: one rank's 128^3 share).  8 ranks without this flag: c3 plus a \"config4\" object")
    ap.add_argument("--c4-cubes", type=int, default=256, help="global cubes per axis of config 4 (tests use less)")
    ap.add_argument("--c4-steps", type=int, default=20, help="timed steps of the appended config-4 measurement")
    ap.add_argument("--grid-sweep", default="448,480,496,512",
                    help="multi-GPU weak-scaling runs: persistent-grid sizes (SEIGEN_HIP_GRID_BLOCKS) of the launches that "
                         "overlap an exchange to time after the main measurement, reported as halo.grid_blocks_sweep "
                         "(empty string: none)")
    ap.add_argument("--configs", default=None,
                    help="one GPU, default headline: the other single-GPU configurations of BASELINE.json and the reference's "
                         "own benchmark protocol, measured after the headline and reported as \"configs\": a comma-separated list "
                         "of c1, c2, c5, c4_share, c4_share_sponge, ref_strong_2d_N256_P<1..4>_T2 (default: all of them with the default headline, "
                         "else none; \"none\": skip)")
    ap.add_argument("--config-timeout", type=float, default=75.0, help="deadline of each entry of --configs, seconds")
    ap.add_argument("--timeout", type=float, default=900.0,
                    help="seconds after which a rank that has not finished dumps its stacks and exits non-zero")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus, args.timeout)      # never returns

    # a rank stuck in init, in an exchange or in a kernel: stacks of all threads to stderr, then exit(1) -
    # from a watchdog thread of the interpreter's own, independent of the GIL
    import faulthandler
    faulthandler.dump_traceback_later(args.timeout, exit=True)
    if os.environ.get("SEIGEN_BENCH_TEST_HANG"):      # tests/test_host_logic.py: a rank that never comes back
        time.sleep(1e6)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # RCCL over xGMI, one rank per GPU.  SEIGEN_DIST_BACKEND=gloo (+ SEIGEN_HIP_DEVICE) runs the same
    # multi-process path with host-staged halos, e.g. two ranks on one GPU (tests/test_dist_gpu.py).
    backend = os.environ.get("SEIGEN_DIST_BACKEND", "nccl")
    if "WORLD_SIZE" in os.environ:      # launched by torch.distributed.run: a process group even for one rank
        import datetime
        import torch
        import torch.distributed as dist
        dev_index = int(os.environ.get("SEIGEN_HIP_DEVICE", local_rank))
        torch.cuda.set_device(dev_index)
        tmo = datetime.timedelta(seconds=max(60.0, args.timeout))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.workload == "c4" and world not in (1, 4, 8) and args.c4_cubes == 256:
        if rank == 0:
            print("bench.py: config 4 (256^3 cubes, 676 GB of fields) needs 4 or 8 GPUs (or 1 for one rank's share)", file=sys.stderr)
        sys.exit(2)

    import seigen_amd
    import seigen_amd.helpers as helpers
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None
    comm = Comm(dist, backend, world)
    P = args.degree
    esz = 4 if args.dtype == "f32" else 8        # bytes per stored value

    workload = args.workload or "c3"
    if workload == "c4":
        steps = args.steps if args.steps is not None else 30
        elastic, grid, gn, wname, scaling = build_config4(args, rank, world, steps + args.warmup + 1)
        args.steps = steps
    else:
        elastic, grid, gn, wname, scaling = build_config3(args, rank, world)
    blk = elastic.block
    n = args.n
    m = measure(elastic, args, comm, args.steps, args.warmup)
    args.steps = m["steps"]
    elapsed, c0, c1, ranks_reporting = m["elapsed"], m["c0"], m["c1"], m["ranks"]

    d = 3
    nodes = blk.ncells * blk.nd
    dofs_per_gpu = blk.u_dofs + blk.s_dofs
    total_dofs = int(round(sum(comm.gather(dofs_per_gpu))))
    value = total_dofs * args.steps / elapsed / 1e6

    # kernels as rocprofv3 names them, from the library; per kernel the device time of the timed region
    kern, ms, nst = kernel_table(blk, c0, c1, esz, region=3 if world > 1 else 0)      # 3 = SG_REGION_FIRST
    assert nst == args.steps, (nst, args.steps)
    mfma = any("mfma_stage" in kname for kname in kern)
    dom = max(kern, key=lambda k: kern[k]["ms"])
    # HBM-side bytes per launch of that kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE, calibrated with
    # tools/calib_fetch.hip) - measured separately (tools/profile_config3.sh -> profiles/<round>/config3_traffic.json)
    # and quoted ONLY if that file was made with the kernels that are running now (digest of seigen_amd/csrc)
    traffic = None
    traffic_src = None
    if world == 1 and workload == "c3" and n == 64 and P == 4 and args.dtype == "f64":
        digest = csrc_digest()
        for tj in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*", "config3_traffic.json"))):
            try:
                tdoc = json.load(open(tj))
                t = tdoc["kernels"].get(dom, {}).get("bytes")
                if t and tdoc.get("csrc_sha256") == digest:
                    traffic, traffic_src = t, os.path.relpath(tj, ROOT)
                elif t:
                    traffic_src = "stale: %s was measured with other kernel sources (csrc digest %s..., running %s...)" % (
                        os.path.relpath(tj, ROOT), str(tdoc.get("csrc_sha256"))[:12], digest[:12])
            except (OSError, ValueError, KeyError):
                pass
    k = kern[dom]
    roof = {"bound": "hbm", "kernel": dom,
            # 9-component accounting of what this launch has to read and write (stage_accounting)
            "achieved": k["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k["gbs"] / HBM_PEAK_GBS,
            # what the kernel moves at best in symmetric-stress mode (6-component stress)
            "achieved_physical": k["gbs_phys"], "frac_physical": k["gbs_phys"] / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_ratio": (traffic / k["bytes_phys_per_launch"]) if traffic else None,
            "traffic_ratio_algorithmic": (traffic / k["bytes_per_launch"]) if traffic else None,
            "algorithmic_bytes_per_launch": k["bytes_per_launch"],
            "physical_bytes_per_launch": k["bytes_phys_per_launch"],
            "avg_launch_ms": k["avg_ms"],
            "kernels": {kk: {"avg_ms": v["avg_ms"], "launches": v["launches"], "algorithmic_GBps": v["gbs"],
                             "physical_GBps": v["gbs_phys"]} for kk, v in kern.items()},
            "bytes_per_dof_update": bytes_per_dof_update(3, esz),
            "whole_step_algorithmic_GBps": dofs_per_gpu * bytes_per_dof_update(3, esz) * args.steps / (elapsed * 1e9),
            "stage_avg_ms": [ms[i] / max(nst, 1) for i in range(6)]}
    if mfma:
        # second view of the same kernel: the dense element-local products on the FP64 matrix pipe.
        # Algorithmic flop per cell and launch: 2 * (9 nd^2 + 12 nd nf), F and G alike.
        nf = (P + 1) * (P + 2) // 2
        flop = 2.0 * (9 * blk.nd ** 2 + 12 * blk.nd * nf) * blk.ncells
        tf = flop / (k["avg_ms"] * 1e-3) / 1e12
        if args.dtype == "f64":
            roof["mfma"] = {"achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                            "sustained_measured": FP64_MFMA_SUSTAINED_TFLOPS, "frac_of_sustained": tf / FP64_MFMA_SUSTAINED_TFLOPS,
                            "algorithmic_flop_per_launch": flop}
        else:   # v_mfma_f32_16x16x4_f32: 157.3 TFLOP/s (/opt/skills/guides/MI355X_MICROARCH.md)
            roof["mfma"] = {"achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "algorithmic_flop_per_launch": flop}

    # halo layer, per rank (lists over ranks): device time of the trace packs, bytes handed to the
    # transport, and the time the launch stream (RCCL) or the host (host-staged) waited for traces
    halo = halo_block(elastic, m, comm, backend) if world > 1 else None
    own = m["own_ms_per_step"]
    out = {
        "metric": "M DoF-updates/sec, 3D elastic P=%d" % P + (" (FP32 second mode)" if args.dtype == "f32" else ""),
        "value": value, "unit": "M DoF-updates/s",
        "n_gpus": ranks_reporting, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": wname, "cells": int(blk.ncells * world) if workload == "c3" else int(6 * gn[0] * gn[1] * gn[2]),
                   "dofs": int(total_dofs), "block_grid": list(grid), "dt": elastic.dt},
        "roofline": roof,
        "timed_region_s": elapsed,
        "rank_ms_per_step": {"min": min(own), "max": max(own), "per_rank": own},
    }
    if halo is not None:
        out["halo"] = halo

    # ---- extras.  The headline record above is complete; nothing below may lose it.  Each extra runs under its own
    # deadline: when it expires with a rank still inside the extra, rank 0 prints the headline with an error note in place
    # of the extra and the job ends THERE.  Multi-GPU extras that time out mean a hang in an exchange: stacks are dumped
    # and the exit code is 3 so that the driver sees it; a single-GPU extra that is merely slow (host-side set-up, the CPU
    # baselines) ends the job with the headline and exit code 0.  An exception inside an extra degrades to the same note.
    import threading
    print_lock = threading.Lock()
    printed = [False]

    def finish(record):
        with print_lock:
            if printed[0]:
                return
            printed[0] = True
            if rank == 0:
                print(json.dumps(record))
                sys.stdout.flush()

    class ExtraDeadline(object):
        def __init__(self, key, seconds, holder):
            self.key, self.holder = key, holder
            self.timer = threading.Timer(seconds, self.expire)
            self.timer.daemon = True
            self.done = self.expired = False

        def expire(self):
            with print_lock:
                if self.done:       # the extra finished while the timer was firing
                    return
                self.expired = True
            # from here on the process ends in this thread, whatever happens: __exit__ sleeps once `expired` is set,
            # so nothing between that and os._exit may be able to raise past the finally
            try:
                try:                # a snapshot the main thread cannot change under us ...
                    note = json.loads(json.dumps(out, default=str))
                except Exception:   # noqa: BLE001 ... or, if it is changing it right now, the top level as it stands
                    note = dict(out)
                target = note if self.holder is None else note.setdefault(self.holder, {})
                target[self.key] = {"error": "timed out after the headline measurement; headline unaffected"}
                if world > 1:
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                finish(note)
            finally:
                os._exit(3 if world > 1 else 0)

        def __enter__(self):
            self.timer.start()
            return self

        def __exit__(self, *exc):
            with print_lock:
                self.done = True
                late = self.expired
            self.timer.cancel()
            if late:                # the deadline fired first: its thread prints the record and ends the process
                time.sleep(1e6)
            return False

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(P)
    extra_budget = max(60.0, min(420.0, 0.4 * args.timeout))
    want_sweep = world > 1 and workload == "c3" and args.grid_sweep and "SEIGEN_HIP_GRID_BLOCKS" not in os.environ
    want_c4 = args.workload is None and world == 8 and P == 4 and not os.environ.get("SEIGEN_BENCH_NO_C4")
    # one GPU, the default headline: every other single-GPU configuration of BASELINE.json and the reference's own
    # benchmark protocol, driver-timed in the same job (--configs none: skip)
    # (--configs given explicitly: whatever the headline's size is - tests use a small one)
    default_headline = n == 64 and P == 4 and args.dtype == "f64"
    if args.configs is None:
        args.configs = ("c1,c2,c5,c4_share,c4_share_sponge,ref_strong_2d_N256_P4_T2,ref_strong_2d_N256_P1_T2,"
                        "ref_strong_2d_N256_P2_T2,ref_strong_2d_N256_P3_T2") if default_headline else "none"
    want_configs = world == 1 and args.workload is None and args.configs != "none"
    if want_sweep or want_c4 or want_configs:
        faulthandler.cancel_dump_traceback_later()
        elastic._exchanger = None
        blk.close()
        del elastic, blk
        elastic = blk = None

    if want_configs:
        out["configs"] = {}
        threads = (out.get("cpu_baseline") or {}).get("cores")
        keys = [c.strip() for c in args.configs.split(",") if c.strip()]
        for key in keys:
            try:
                with ExtraDeadline(key, args.config_timeout, "configs"):
                    out["configs"][key] = secondary_config(key, with_cpu=not args.no_cpu_baseline, threads=threads,
                                                           keep_c4="c4_share_sponge" in keys[keys.index(key) + 1:keys.index(key) + 2])
            except Exception as e:      # noqa: BLE001 - anything here must not cost the headline
                out["configs"][key] = {"error": repr(e)}

    # multi-GPU, weak-scaling workload: how the step time depends on the share of block slots the launches that run
    # beside an exchange leave to RCCL's kernels (default 15/16; it was tuned on ONE device against a self-send)
    if want_sweep:
        sweep = {}
        try:
            with ExtraDeadline("grid_blocks_sweep_ms_per_step", extra_budget, "halo"):
                for gb in [v for v in args.grid_sweep.split(",") if v.strip()]:
                    os.environ["SEIGEN_HIP_GRID_BLOCKS"] = gb.strip()
                    el_s, _, _, _, _ = build_config3(args, rank, world)
                    ms_s = measure(el_s, args, comm, 30, 3)
                    sweep[gb.strip()] = ms_s["elapsed"] / ms_s["steps"] * 1e3
                    el_s._exchanger = None
                    el_s.block.close()
                    del el_s
            out["halo"]["grid_blocks_sweep_ms_per_step"] = sweep
        except Exception as e:      # noqa: BLE001 - anything here must not cost the headline
            out["halo"]["grid_blocks_sweep_ms_per_step"] = {"error": repr(e), "partial": sweep}
        os.environ.pop("SEIGEN_HIP_GRID_BLOCKS", None)

    # 8 ranks, no explicit workload: BASELINE config 4 in the same job (its own mesh, its own barriers)
    if want_c4:
        try:
            with ExtraDeadline("config4", extra_budget, None):
                el4, grid4, gn4, wname4, _ = build_config4(args, rank, world, args.c4_steps + 3 + 1)
                m4 = measure(el4, args, comm, args.c4_steps, 3)
                dofs4 = int(round(sum(comm.gather(el4.block.u_dofs + el4.block.s_dofs))))
                own4 = m4["own_ms_per_step"]
                c4 = {"workload": wname4, "value": dofs4 * m4["steps"] / m4["elapsed"] / 1e6, "unit": "M DoF-updates/s",
                      "ms_per_step": m4["elapsed"] / m4["steps"] * 1e3, "steps": m4["steps"], "warmup": 3,
                      "scaling": "strong (fixed global size)", "block_grid": list(grid4), "dofs": dofs4,
                      "block_cubes": list(el4.mesh.partition.n), "dt": el4.dt,
                      "rank_ms_per_step": {"min": min(own4), "max": max(own4), "per_rank": own4},
                      "halo": halo_block(el4, m4, comm, backend)}
            out["config4"] = c4
        except Exception as e:      # noqa: BLE001
            out["config4"] = {"error": repr(e)}
    finish(out)
    faulthandler.cancel_dump_traceback_later()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
