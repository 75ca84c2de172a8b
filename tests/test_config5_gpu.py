"""BASELINE config 5 as a workload: Marmousi material (seigen/data/marmhard.dat through the lookup
of seigen/marmousi.py:4-14) driving per-cell lambda, mu - and per-cell density - on the
383 x 121-square P3 mesh of seigen/marmousi.py:16-21.

The reference defines no semantics for discontinuous material; the build's are: each
cell scales its own g by its own (lambda_K, mu_K) and its own velocity update by its own rho_K.
The oracle implements the same rule (oracle/forms.py apply_G, oracle/lf4.py), so:

  * crops of the model (24 x 12 squares: around the largest Vp contrast, at the surface, at the
    bottom edge) run 20 steps on the HIP path and on the oracle with the same per-cell arrays and
    must agree to 1e-10;
  * at the full size the run is checked through exact time reversal (tests/test_fullsize_gpu.py)
    with the heterogeneous lambda, mu AND a per-cell Gardner density in the physical update.
"""
import numpy as np
import pytest

from oracle import mesh as omesh
from oracle.lf4 import OracleLF4
from tests.util import rel_err

pytestmark = pytest.mark.gpu

FORWARD = (0, 1, 2, 3, 4, 5)
BACKWARD = (3, 4, 5, 0, 1, 2)


def _quiet():
    import seigen_amd
    import seigen_amd.helpers as helpers
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None


def _crop_origin(data, where, nx, ny):
    """Low corner (in model cells) of an nx x ny crop."""
    from seigen_amd.marmousi import NX, NY
    if where == "contrast":
        # largest jump of Vp between vertically adjacent model cells; rows count down from the surface
        jump = np.abs(np.diff(data, axis=1))
        i, r = np.unravel_index(np.argmax(jump), jump.shape)
        j = NY - 1 - r                       # mesh row whose upper edge carries that jump (rule "fixed")
        i0 = int(np.clip(i - nx // 2, 0, NX - 1 - nx))
        j0 = int(np.clip(j - ny // 2, 0, NY - 1 - ny))
        return i0, j0
    if where == "surface":
        return 180, NY - 1 - ny
    return 40, 0                              # bottom edge: the j = 0 row of the lookup rule


@pytest.mark.parametrize("where,path,density", [
    ("contrast", None, "unit"), ("contrast", "lane", "gardner"), ("surface", None, "gardner"), ("bottom", None, "unit"),
])
def test_marmousi_crop_matches_oracle(gpu, monkeypatch, where, path, density):
    _quiet()
    if path:
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
    from seigen_amd import ElasticLF4, Function, RectangleMesh, cfl_dt
    from seigen_amd.marmousi import H, cell_material, gardner_density, load_model
    data = load_model()
    nx, ny, P, nsteps = 24, 12, 3, 20
    i0, j0 = _crop_origin(data, where, nx, ny)
    origin = (i0 * H, j0 * H)
    mesh = RectangleMesh(nx, ny, nx * H, ny * H)
    mesh.origin = origin
    el = ElasticLF4.create(mesh, "DG", P, dimension=2, solver="explicit", output=False)
    rho = gardner_density if density == "gardner" else 1.0
    lam, mu, vp = cell_material(el.U, data, density=rho)
    rho_cells = gardner_density(vp) if density == "gardner" else 1.0
    if where == "contrast":
        assert vp.max() / vp.min() > 1.5, "the crop must contain a real contrast"
    el.density, el.l, el.mu = rho_cells, lam, mu
    el.density_physical = density == "gardner"
    el.dt = cfl_dt(H, float(vp.max()), 0.05)
    rng = np.random.default_rng(11)
    u0 = rng.uniform(-1, 1, (el.U.ncells, el.U.nd, 2))
    s0 = rng.uniform(-1, 1, (el.S.ncells, el.S.nd, 2, 2))
    s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2)) * float(lam.mean())      # stress-sized, symmetric
    el.u0.assign(Function(el.U).assign(u0))
    el.s0.assign(Function(el.S).assign(s0))
    u1, s1 = el.run(nsteps * el.dt * (1 + 1e-9))
    assert el.block.counters()["steps"] == nsteps

    m = omesh.structured(2, (nx, ny), (nx * H, ny * H), origin=origin)
    np.testing.assert_allclose(m.node_coords(P), el.U.node_coords(), rtol=0, atol=1e-9)
    orc = OracleLF4(m, P)
    orc.dt, orc.l, orc.mu = el.dt, lam, mu
    orc.density, orc.density_physical = rho_cells, el.density_physical
    orc.u0, orc.s0 = u0.copy(), s0.copy()
    for k in range(nsteps):
        orc.step((k + 1) * orc.dt)
    assert rel_err(u1.dat.data_cells, orc.u1) < 1e-10
    assert rel_err(s1.dat.data_cells, orc.s1) < 1e-10
    # heterogeneity matters: the same run with the mean material differs at O(1)
    assert rel_err(orc.u1, u0) > 1e-3


def test_per_cell_density_both_updates(gpu):
    """sg_set_density on every kernel family: reference update rho_K*u0 + ..., physical u0 + (...)/rho_K."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    for dim, degree, n, L in ((1, 2, (9,), (2.0,)), (2, 3, (4, 3), (1.0, 1.0)), (3, 2, (2, 2, 3), (1.0, 1.0, 1.0)),
                              (3, 4, (2, 2, 2), (1.0, 1.0, 1.0))):
        for physical in (False, True):
            h = [L[a] / n[a] for a in range(dim)]
            blk = HipBlock(dim, degree, n, h, [0.0] * dim)
            m = omesh.structured(dim, n, L)
            orc = OracleLF4(m, degree)
            rng = np.random.default_rng(3)
            rho = rng.uniform(0.6, 2.5, m.ncells)
            orc.dt, orc.l, orc.mu = 1e-3, 0.5, 0.25
            orc.density, orc.density_physical = rho, physical
            orc.u0 = rng.uniform(-1, 1, blk.field_shape(_lib.FIELD_U))
            orc.s0 = rng.uniform(-1, 1, blk.field_shape(_lib.FIELD_S))
            blk.set_params(1.0, orc.dt, orc.l, orc.mu)
            blk.set_density(rho, physical)
            blk.set_field(_lib.FIELD_U, orc.u0)
            blk.set_field(_lib.FIELD_S, orc.s0)
            blk.step(3)
            for k in range(3):
                orc.step((k + 1) * orc.dt)
            assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 1e-10, (dim, degree, physical)
            assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 1e-10, (dim, degree, physical)
            # scalar density in the physical update, and sg_set_params resetting to the reference update
            blk.set_density(1.7, True)
            orc2 = OracleLF4(m, degree)
            orc2.dt, orc2.l, orc2.mu, orc2.density, orc2.density_physical = orc.dt, 0.5, 0.25, 1.7, True
            orc2.u0, orc2.s0 = orc.u1.copy(), orc.s1.copy()
            blk.step(1)
            orc2.step(orc.dt)
            assert rel_err(blk.get_field(_lib.FIELD_U), orc2.u1) < 1e-10
            blk.close()


def test_config5_full_size_time_reversal_heterogeneous(gpu):
    """383 x 121 squares, P3, Marmousi lambda/mu and Gardner density per cell (physical update):
    20 steps forward, 20 back, initial state recovered to round-off."""
    _quiet()
    from seigen_amd import ElasticLF4, RectangleMesh, _lib, cfl_dt
    from seigen_amd.marmousi import NX, NY, H, cell_material, gardner_density
    mesh = RectangleMesh(NX - 1, NY - 1, (NX - 1) * H, (NY - 1) * H)           # seigen/marmousi.py:16-21
    el = ElasticLF4.create(mesh, "DG", 3, dimension=2, solver="explicit", output=False)
    lam, mu, vp = cell_material(el.U, density=gardner_density)
    assert 1500.0 <= vp.min() < 1600.0 and vp.max() == 5500.0 and len(np.unique(vp)) > 100
    el.density, el.density_physical, el.l, el.mu = gardner_density(vp), True, lam, mu
    el.dt = dt = cfl_dt(H, float(vp.max()), 0.05)
    el.setup()
    blk = el.block
    blk.set_source([], None)
    X = el.U.node_coords() / 1000.0
    k = np.array([[2.0, 3.0], [3.5, 1.5]])
    u0 = np.stack([np.sin(X @ k[i]) for i in range(2)], axis=-1)
    s0 = np.zeros(X.shape[:-1] + (2, 2))
    for i in range(2):
        for j in range(i, 2):
            s0[..., i, j] = s0[..., j, i] = 1e7 * np.cos(X @ k[(i + j) % 2] + i - j)
    blk.set_field(_lib.FIELD_U, u0)
    blk.set_field(_lib.FIELD_S, s0)
    K = 20
    for _ in range(K):
        for st in FORWARD:
            blk.run_stage(st)
        blk.end_step()
    u_mid = blk.get_field(_lib.FIELD_U)
    assert np.abs(u_mid - u0).max() > 1e-3, "the forward steps must change the state"
    blk.set_params(1.0, -dt, lam, mu)
    blk.set_density(gardner_density(vp), True)
    for _ in range(K):
        for st in BACKWARD:
            blk.run_stage(st)
        blk.end_step()
    du = np.abs(blk.get_field(_lib.FIELD_U) - u0).max()
    ds = np.abs(blk.get_field(_lib.FIELD_S) - s0).max() / 1e7
    assert du < 1e-10 and ds < 1e-10, (du, ds)


@pytest.mark.parametrize("case", ["stiffer", "denser"])
def test_two_layer_reflection_and_transmission_1d(gpu, case):
    """The heterogeneous extension against theory on the HIP path (oracle counterpart: tests/test_oracle_pins.py): a
    pulse in a 1-D bar meets an interface between two media (per-cell lambda, mu, density in the physical update) and
    splits into a reflected and a transmitted pulse with amplitudes (Z1 - Z2) / (Z1 + Z2) and 2 Z1 / (Z1 + Z2)."""
    _quiet()
    import math
    from seigen_amd import ElasticLF4, Function, IntervalMesh
    n, L, P = 400, 4.0, 3
    el = ElasticLF4.create(IntervalMesh(n, L), "DG", P, dimension=1, solver="explicit", output=False)
    X = el.U.node_coords()[..., 0]
    xi = 2.0
    right = X.mean(axis=1) > xi
    rho1, M1 = 1.0, 1.0
    rho2, M2 = (1.0, 4.0) if case == "stiffer" else (4.0, 1.0)
    c1, c2 = math.sqrt(M1 / rho1), math.sqrt(M2 / rho2)
    Z1, Z2 = rho1 * c1, rho2 * c2
    el.l, el.mu = np.where(right, M2 / 2.0, M1 / 2.0), np.where(right, M2 / 4.0, M1 / 4.0)
    el.density, el.density_physical, el.dt = np.where(right, rho2, rho1), True, 0.0005
    g = lambda x: np.exp(-50.0 * (x - 1.0) ** 2)
    el.u0.assign(Function(el.U).assign(g(X)[..., None]))
    el.s0.assign(Function(el.S).assign((-Z1 * g(X))[..., None, None]))
    T = 1.5
    u1, s1 = el.run(T)
    R, Tc = (Z1 - Z2) / (Z1 + Z2), 2 * Z1 / (Z1 + Z2)
    exact = np.where(X < xi, g(X - c1 * T) + R * g(2 * xi - X - c1 * T), Tc * g(xi + (X - xi) * c1 / c2 - c1 * T))
    assert np.abs(u1.dat.data_cells[..., 0] - exact).max() < 1e-3, case
