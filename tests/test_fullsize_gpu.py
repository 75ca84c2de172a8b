"""BASELINE's full-size configurations on the GPU, checked through size-independent properties
(the oracle cannot run these sizes in seconds):

* the analytic eigenmode (tests/eigenmode/eigenmode_3d.py:30-40) sampled all over the mesh,
* exact time reversal of the LF4 update: with central fluxes and no sponge the scheme
  (seigen/elastic.py:291-304, :340-352) is reversible; undoing a step is the stress update run
  with -dt followed by the velocity update with -dt, so K steps forward and K steps backward must
  return the initial state to round-off.  Every launch of the step (plain and fused kernels,
  interior and domain-boundary facets) takes part at the full size,
* polynomial reproduction: g of a linear velocity field is the constant Hooke stress everywhere,
  domain boundary included (own-trace boundary flux, elastic.py:216).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FORWARD = (0, 1, 2, 3, 4, 5)
BACKWARD = (3, 4, 5, 0, 1, 2)     # with dt negated: undo the stress update, then the velocity update


def _quiet():
    import seigen_amd
    import seigen_amd.helpers as helpers
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None


def _sample_ranges(ncells, per=96, count=7):
    """(cell0, ncells) pieces spread over the whole block, first and last cells included."""
    starts = np.linspace(0, ncells - per, count).astype(np.int64)
    return [(int(s), per) for s in starts]


def _advance(blk, order, nsteps):
    for _ in range(nsteps):
        for stage in order:
            blk.run_stage(stage)
        blk.end_step()


def test_config3_eigenmode_and_time_reversal_full_size(gpu):
    """BASELINE config 3: 64^3 cubes x 6 tets, P4 (1 572 864 cells, 660.6 M DoF)."""
    _quiet()
    import bench
    from seigen_amd import ElasticLF4, BoxMesh, _lib
    from seigen_amd.functionspace import block_config
    import ctypes as C
    n, P = 64, 4
    mesh = BoxMesh(n, n, n, 1.0, 1.0, 1.0)
    el = ElasticLF4.create(mesh, "DG", P, dimension=3, solver="explicit", output=False)
    el.density, el.mu, el.l = 1.0, 0.25, 0.5
    el.dt = dt = 0.5 * (1.0 / n) / 2 ** (P - 1)
    bench.fill_initial_condition(el, dt)
    el.setup()
    blk = el.block
    blk.set_source([], None)
    assert blk.ncells == 6 * n ** 3
    # node coordinates of the sampled cells: whole z-layers of cubes, as bench.py builds them
    lib = _lib.load()
    layer_cells = n * n * 6
    layers = (0, 21, 42, 63)

    def layer_coords(k):
        cfg = block_config(mesh, P)
        cfg.n[2] = 1
        cfg.origin[2] = k * mesh.h[2]
        X = np.empty((layer_cells, blk.nd, 3))
        _lib.check(lib.sg_block_node_coords(C.byref(cfg), P, X.ctypes.data, X.nbytes))
        return X

    pieces = [(k, c0, 192) for k in layers for c0 in (0, layer_cells // 2 - 96, layer_cells - 192)]
    u_ic = {p: blk.get_field_range(_lib.FIELD_U, p[0] * layer_cells + p[1], p[2]) for p in pieces}
    s_ic = {p: blk.get_field_range(_lib.FIELD_S, p[0] * layer_cells + p[1], p[2]) for p in pieces}

    K = 4
    _advance(blk, FORWARD, K)
    blk.sync()
    # analytic solution: u at t = K dt, s at t = K dt + dt/2 (staggered, eigenmode_3d.py:32-38)
    worst_u = worst_s = 0.0
    for p in pieces:
        X = layer_coords(p[0])[p[1]:p[1] + p[2]]
        ue, se = bench.eigenmode3d_fields(X, K * dt, K * dt + dt / 2)
        worst_u = max(worst_u, np.abs(blk.get_field_range(_lib.FIELD_U, p[0] * layer_cells + p[1], p[2]) - ue).max())
        worst_s = max(worst_s, np.abs(blk.get_field_range(_lib.FIELD_S, p[0] * layer_cells + p[1], p[2]) - se).max())
    # P4 at h = 1/64: interpolation-level errors (fields are O(1)); any indexing slip would give O(1)
    assert worst_u < 1e-7 and worst_s < 1e-7, (worst_u, worst_s)

    blk.set_params(1.0, -dt, 0.5, 0.25)
    _advance(blk, BACKWARD, K)
    blk.sync()
    for p in pieces:
        du = np.abs(blk.get_field_range(_lib.FIELD_U, p[0] * layer_cells + p[1], p[2]) - u_ic[p]).max()
        ds = np.abs(blk.get_field_range(_lib.FIELD_S, p[0] * layer_cells + p[1], p[2]) - s_ic[p]).max()
        assert du < 1e-12 and ds < 1e-12, (p, du, ds)


@pytest.mark.parametrize("dim,degree,n,steps", [
    (2, 2, (512, 512), 6),        # BASELINE config 2's mesh and order (2-D MFMA tile kernels)
    (2, 3, (383, 121), 6),        # BASELINE config 5's mesh and order (2-D MFMA tile kernels)
    (3, 3, (48, 40, 36), 3),      # MFMA P3, ragged sizes (layout padding in x)
    (3, 2, (48, 48, 48), 3),      # 3-D P2 (MFMA kernels on 4-row tiles)
])
def test_time_reversal_and_linear_reproduction(gpu, dim, degree, n, steps):
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    h = tuple(1.0 / max(n) for _ in n)
    blk = HipBlock(dim, degree, n, h, (0.0,) * dim)
    lam, mu = 0.5, 0.25
    dt = 0.25 * h[0] / degree ** 2
    blk.set_params(1.0, dt, lam, mu)
    X = blk.node_coords(degree)                       # [cells, nd, dim]
    # ---- g of a linear velocity field = constant Hooke stress, everywhere
    A = np.arange(1, dim * dim + 1, dtype=np.float64).reshape(dim, dim) / 7.0 - 0.4
    u_lin = np.einsum("ij,cnj->cni", A, X) + 0.3
    blk.set_field(_lib.FIELD_U, u_lin)
    blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
    hooke = lam * np.trace(A) * np.eye(dim) + mu * (A + A.T)
    for c0, nc in _sample_ranges(blk.ncells):
        got = blk.get_field_range(_lib.FIELD_SH, c0, nc)
        assert np.abs(got - hooke).max() < 1e-10 * max(n), (c0, np.abs(got - hooke).max())
    # ---- time reversal from smooth, non-trivial data
    rng = np.random.default_rng(5)
    k = rng.uniform(1.0, 4.0, size=(dim, dim))
    u0 = np.stack([np.sin(X @ k[i]) for i in range(dim)], axis=-1)
    s0 = np.zeros(X.shape[:-1] + (dim, dim))
    for i in range(dim):
        for j in range(i, dim):
            s0[..., i, j] = s0[..., j, i] = np.cos(X @ k[(i + j) % dim] + i - j)
    blk.set_field(_lib.FIELD_U, u0)
    blk.set_field(_lib.FIELD_S, s0)
    _advance(blk, FORWARD, steps)
    moved = np.abs(blk.get_field_range(_lib.FIELD_U, 0, 64) - u0[:64]).max()
    assert moved > 1e-6, "the forward steps must change the state"
    blk.set_params(1.0, -dt, lam, mu)
    _advance(blk, BACKWARD, steps)
    for c0, nc in _sample_ranges(blk.ncells):
        du = np.abs(blk.get_field_range(_lib.FIELD_U, c0, nc) - u0[c0:c0 + nc]).max()
        ds = np.abs(blk.get_field_range(_lib.FIELD_S, c0, nc) - s0[c0:c0 + nc]).max()
        assert du < 1e-11 and ds < 1e-11, (c0, du, ds)
    blk.close()


@pytest.mark.parametrize("config", ["c2", "c5"])
def test_full_size_2d_workloads_tile_vs_generic(gpu, monkeypatch, config):
    """BASELINE configs 2 and 5 as workloads at full size - 512 x 512 squares, P2, DG4 sponge and box-Ricker source
    (tests/explosive_source/explosive_source_lf4.py:17-45); Marmousi 383 x 121, P3, per-cell lambda / mu and a Ricker
    source - 20 LF4 steps on the production path (2-D MFMA tile kernels) against the independently written generic
    kernel family (itself checked against the oracle at small sizes): every field, everywhere, to 1e-10."""
    from seigen_amd import _lib
    import seigen_amd.elastic
    import seigen_amd.harness.explosive_source as hx
    from seigen_amd.harness import baseline_configs as bc     # the set-ups bench.py's "configs" object measures
    monkeypatch.setattr(seigen_amd.elastic, "log", lambda s: None)
    monkeypatch.setattr(hx, "log", lambda s: None)
    steps, out = 20, {}
    for path in ("generic", "tile"):
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        elastic, _ = (bc.config2 if config == "c2" else bc.config5)(steps)
        nodes, _, _ = elastic._source_table([elastic.dt * (k + 1) for k in range(steps)])
        assert len(nodes) > 0
        blk = elastic.block
        blk.step(steps)
        out[path] = tuple(blk.get_field(f) for f in (_lib.FIELD_U, _lib.FIELD_S, _lib.FIELD_UH, _lib.FIELD_SH))
        blk.close()
    assert np.isfinite(out["generic"][1]).all() and np.abs(out["generic"][1]).max() > 0
    for a, b in zip(out["tile"], out["generic"]):
        scale = max(np.abs(b).max(), 1e-300)
        assert np.abs(a - b).max() / scale < 1e-10


def test_config3_full_size_mfma_vs_generic(gpu, monkeypatch):
    """BASELINE config 3 at full size (64^3 cubes x 6 tets, P4): three LF4 steps of the eigenmode on the production
    path (MFMA kernels, interleaved layout, symmetric-stress storage) against the independently written generic
    kernel family (host layout, full tensor), sampled over the whole block.  The state (u, s) must agree to 1e-11;
    the stage fields left behind are derivatives of it - sh1 = G(u1), utemp = F(G(u1)) - and carry the operators'
    round-off amplification, eps (P^2/h)^k with P^2/h ~ 1e3..1e4 here: 1e-9 and 1e-7 of their scale."""
    _quiet()
    import bench
    from seigen_amd import ElasticLF4, BoxMesh, _lib
    n, P, steps = 64, 4, 3
    samples = {}
    for path in ("generic", "mfma"):
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        mesh = BoxMesh(n, n, n, 1.0, 1.0, 1.0)
        el = ElasticLF4.create(mesh, "DG", P, dimension=3, solver="explicit", output=False)
        el.density, el.mu, el.l = 1.0, 0.25, 0.5
        el.dt = 0.5 * (1.0 / n) / 2 ** (P - 1)
        bench.fill_initial_condition(el, el.dt)
        el.setup()
        blk = el.block
        blk.set_source([], None)
        blk.step(steps)
        samples[path] = [(f, c0, blk.get_field_range(f, c0, nc)) for f in (_lib.FIELD_U, _lib.FIELD_S, _lib.FIELD_UH, _lib.FIELD_SH)
                         for c0, nc in _sample_ranges(blk.ncells, per=192, count=9)]
        blk.close()
        del el
    tol = {_lib.FIELD_U: 1e-11, _lib.FIELD_S: 1e-11, _lib.FIELD_SH: 1e-9, _lib.FIELD_UH: 1e-7}
    scale = {f: max(np.abs(b).max() for g, _, b in samples["generic"] if g == f) for f in tol}   # per field, over all samples
    assert all(v > 1e-4 for v in scale.values()), scale      # (UH holds w = dt u1 + dt^3/24 utemp: of the size of dt)
    for (f, c0, a), (_, _, b) in zip(samples["mfma"], samples["generic"]):
        assert np.abs(a - b).max() / scale[f] < tol[f], (f, c0, np.abs(a - b).max() / scale[f])


@pytest.mark.parametrize("P,n", [(3, 48), (4, 40)])
def test_config3_on_hexahedra_full_size_matrix_vs_generic(gpu, monkeypatch, P, n):
    """Config 3's eigenmode on HEXAHEDRA at the sizes the bench quotes (DQ_3 48^3, DQ_4 40^3): three LF4 steps on
    the production path (kernels_hexm.hip: plane by plane, x lines on the matrix pipe, gw = 16 layout, symmetric storage)
    against the independently written thread-per-node generic kernels (host layout, full tensor), sampled over the whole
    block, and against the analytic mode the run started from (eigenmode_3d.py:30-40).  Tolerances as for the tetrahedra:
    the stage fields carry the operators' round-off amplification."""
    _quiet()
    from seigen_amd import Function, _lib
    from seigen_amd.harness.eigenmode import Eigenmode3DLF4
    import seigen_amd.harness.eigenmode as he
    monkeypatch.setattr(he, "log", lambda s: None)
    steps, samples, exact = 3, {}, {}
    for path in ("generic", "hexm"):
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        em = Eigenmode3DLF4(n, P, 0.5 * (1.0 / n) / 2.0 ** (P - 1), output=False, hexahedral=True)
        el = em.elastic
        el.u0.assign(Function(el.U).interpolate(em._u(0)))
        el.s0.assign(Function(el.S).interpolate(em._s(el.dt / 2)))
        el.setup()
        blk = el.block
        assert ("hexm_stage" in blk.stage_kernel_name(0)) == (path == "hexm")
        blk.set_source([], None)
        blk.step(steps)
        samples[path] = [(f, c0, blk.get_field_range(f, c0, nc)) for f in (_lib.FIELD_U, _lib.FIELD_S, _lib.FIELD_UH, _lib.FIELD_SH)
                         for c0, nc in _sample_ranges(blk.ncells, per=64, count=9)]
        if path == "hexm":
            ue = Function(el.U).interpolate(em._u(steps * el.dt)).dat.data_cells
            exact = {c0: ue[c0:c0 + nc] for f, c0, a in samples[path] if f == _lib.FIELD_U for nc in (a.shape[0],)}
        blk.close()
        del el, em
    tol = {_lib.FIELD_U: 1e-11, _lib.FIELD_S: 1e-11, _lib.FIELD_SH: 1e-9, _lib.FIELD_UH: 1e-7}
    scale = {f: max(np.abs(b).max() for g, _, b in samples["generic"] if g == f) for f in tol}
    assert all(v > 1e-4 for v in scale.values()), scale      # (UH holds w = dt u1 + dt^3/24 utemp: of the size of dt)
    for (f, c0, a), (_, _, b) in zip(samples["hexm"], samples["generic"]):
        assert np.abs(a - b).max() / scale[f] < tol[f], (f, c0, np.abs(a - b).max() / scale[f])
        if f == _lib.FIELD_U:       # three steps of a degree-3 / 4 scheme on this mesh stay within 1e-6 of the analytic mode
            assert np.abs(a - exact[c0]).max() < 1e-6, (c0, np.abs(a - exact[c0]).max())


def test_config4_share_full_size_properties(gpu):
    """One rank's share of BASELINE config 4 (3-D 256^3 on 8 GPUs): a 128^3-cube block, P4, 12.6 M cells, 5.3 G DoF,
    85 GB resident.  g of a linear velocity field = the constant Hooke stress everywhere (sampled), and exact time
    reversal: two LF4 steps forward from (u linear, s = 0), two back, the linear field returns to round-off."""
    import ctypes as C
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    n, P = 128, 4
    h = (1.0 / n,) * 3
    blk = HipBlock(3, P, (n, n, n), h, (0.0, 0.0, 0.0))
    assert blk.ncells == 6 * n ** 3
    lam, mu = 0.5, 0.25
    dt = 0.25 * h[0] / P ** 2
    blk.set_params(1.0, dt, lam, mu)
    A = np.arange(1, 10, dtype=np.float64).reshape(3, 3) / 7.0 - 0.4
    layer = n * n * 6
    lib = _lib.load()

    def layer_coords(k):          # node coordinates of the k-th z-layer of cubes (the library's own numbering)
        cfg = _lib.SgConfig()
        cfg.dim, cfg.degree = 3, P
        for a in range(3):
            cfg.n[a], cfg.h[a], cfg.origin[a] = n, h[a], 0.0
        cfg.n[2] = 1
        cfg.origin[2] = k * h[2]
        X = np.empty((layer, blk.nd, 3))
        _lib.check(lib.sg_block_node_coords(C.byref(cfg), P, X.ctypes.data, X.nbytes))
        return X

    u_of = lambda X: np.einsum("ij,cnj->cni", A, X) + 0.3
    for k in range(n):
        blk.set_field_range(_lib.FIELD_U, k * layer, u_of(layer_coords(k)))
    blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
    hooke = lam * np.trace(A) * np.eye(3) + mu * (A + A.T)
    for c0, nc in _sample_ranges(blk.ncells, per=192, count=9):
        got = blk.get_field_range(_lib.FIELD_SH, c0, nc)
        assert np.abs(got - hooke).max() < 1e-10 * n, (c0, np.abs(got - hooke).max())
    _advance(blk, FORWARD, 2)
    k_mid = n // 2
    moved = np.abs(blk.get_field_range(_lib.FIELD_U, k_mid * layer, 192) - u_of(layer_coords(k_mid))[:192]).max()
    assert moved > 1e-9, "the forward steps must change the state"
    blk.set_params(1.0, -dt, lam, mu)
    _advance(blk, BACKWARD, 2)
    for k in (0, 37, k_mid, n - 1):
        ref = u_of(layer_coords(k))
        for c0 in (0, layer // 2, layer - 192):
            du = np.abs(blk.get_field_range(_lib.FIELD_U, k * layer + c0, 192) - ref[c0:c0 + 192]).max()
            ds = np.abs(blk.get_field_range(_lib.FIELD_S, k * layer + c0, 192)).max()
            assert du < 1e-11 and ds < 1e-11, (k, c0, du, ds)
    blk.close()
