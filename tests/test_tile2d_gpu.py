"""GPU parity of the 2-D MFMA tile kernels (seigen_amd/csrc/kernels_tile2d.hip), forced through
SEIGEN_HIP_PATH=tile on meshes small enough for the oracle: operators and whole LF4 steps against the
oracle (seigen/elastic.py:204-219, :340-352), ragged widths (16-square groups that straddle rows),
both diagonals, sponge / source / per-cell material and density, symmetric and full-tensor stress, and
multi-block runs (packed remote traces) bitwise equal to the single block."""
import numpy as np
import pytest

from oracle.forms import ElasticOperators
from oracle.lf4 import OracleLF4
from tests.util import oracle_mesh, rel_err, seeded

pytestmark = pytest.mark.gpu

TOL = 1e-11

CASES = [
    # degree, n, L, diagonal
    (1, (4, 4), (1.0, 1.0), "left"),
    (1, (37, 3), (2.0, 0.5), "right"),
    (2, (16, 2), (1.0, 1.0), "left"),
    (2, (7, 9), (1.0, 1.5), "right"),
    (3, (5, 4), (1.0, 1.25), "left"),
    (3, (19, 3), (1.0, 1.0), "right"),
    (4, (3, 4), (1.5, 1.0), "left"),
    (4, (18, 2), (1.0, 1.0), "left"),
]


def make_block(degree, n, L, diagonal, nbr_mask=0):
    from seigen_amd.backend import HipBlock
    h = [L[a] / n[a] for a in range(2)]
    return HipBlock(2, degree, n, h, [0.0, 0.0], diagonal, nbr_mask)


@pytest.fixture
def tile(monkeypatch):
    monkeypatch.setenv("SEIGEN_HIP_PATH", "tile")


@pytest.mark.parametrize("degree,n,L,diagonal", CASES)
def test_tile_apply_F_and_G(gpu, tile, degree, n, L, diagonal):
    from seigen_amd import _lib
    blk = make_block(degree, n, L, diagonal)
    E = ElasticOperators(oracle_mesh(2, n, L, diagonal), degree)
    for symmetric in (True, False):
        T = seeded(blk.field_shape(_lib.FIELD_S), 0)
        if symmetric:
            T = 0.5 * (T + np.swapaxes(T, -1, -2))
        u = seeded(blk.field_shape(_lib.FIELD_U), 1)
        lam, mu = 0.7, 0.3
        blk.set_params(1.0, 0.01, lam, mu)
        blk.set_field(_lib.FIELD_S, T)
        blk.set_field(_lib.FIELD_U, u)
        np.testing.assert_array_equal(blk.get_field(_lib.FIELD_S), T)
        blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
        assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(T, u)) < TOL
        blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
        assert rel_err(blk.get_field(_lib.FIELD_SH), E.apply_G(u, lam, mu)) < TOL


@pytest.mark.parametrize("degree,n,L,diagonal", CASES)
def test_tile_full_steps(gpu, tile, degree, n, L, diagonal):
    """Three whole LF4 steps (six fused launches each) against the un-fused oracle, with a density != 1
    (the explicit reference's rho*u0 quirk, elastic.py:341-345)."""
    from seigen_amd import _lib
    blk = make_block(degree, n, L, diagonal)
    orc = OracleLF4(oracle_mesh(2, n, L, diagonal), degree)
    orc.dt = 0.05 * min(L[a] / n[a] for a in range(2)) / degree ** 2
    orc.l, orc.mu, orc.density = 0.5, 0.25, 1.3
    orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 2)
    s0 = seeded(blk.field_shape(_lib.FIELD_S), 3)
    orc.s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
    blk.set_params(orc.density, orc.dt, orc.l, orc.mu)
    blk.set_field(_lib.FIELD_U, orc.u0)
    blk.set_field(_lib.FIELD_S, orc.s0)
    blk.step(3)
    for k in range(3):
        orc.step((k + 1) * orc.dt)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 10 * TOL
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 10 * TOL
    assert rel_err(blk.get_field(_lib.FIELD_UH), orc.dt * orc.u1 + orc.dt ** 3 / 24.0 * orc.last["utemp"]) < 10 * TOL
    assert rel_err(blk.get_field(_lib.FIELD_SH), orc.last["sh1"]) < 10 * TOL
    # a non-symmetric stress uploaded mid-run: the handle leaves symmetric mode and stays exact
    s_now = blk.get_field(_lib.FIELD_S)
    s_now[..., 0, 1] += 0.125
    blk.set_field(_lib.FIELD_S, s_now)
    orc.s0 = s_now.copy()
    orc.step(4 * orc.dt)
    blk.step(1)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 10 * TOL
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 10 * TOL


@pytest.mark.parametrize("degree,n", [(1, (21, 5)), (2, (9, 6)), (3, (18, 4)), (4, (5, 7))])
def test_tile_sponge_source_material_density_vs_generic(gpu, monkeypatch, degree, n):
    """Sponge (DG4 sigma, elastic.py:207-208), sparse time-dependent source (:217-218), per-cell lambda / mu
    and per-cell density in the physical update: the tile kernels against the generic kernel (itself
    checked against the oracle in test_parity_gpu / test_harness_gpu)."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    h = (0.9 / n[0], 1.1 / n[1])
    dt = 0.05 * min(h) / degree ** 2
    res = {}
    for path in ("generic", "tile"):
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        blk = HipBlock(2, degree, n, h, (0.0, 0.0), "left")
        r = np.random.default_rng(5)
        blk.set_params(1.0, dt, r.uniform(0.4, 0.9, blk.ncells), r.uniform(0.2, 0.5, blk.ncells))
        blk.set_density(r.uniform(0.8, 1.6, blk.ncells), physical=True)
        blk.set_absorption(np.where(r.uniform(size=(blk.ncells, 15)) > 0.6, 4.0, 0.0), 4)
        nodes = np.unique(r.integers(0, blk.ncells * blk.nd, size=30))
        sv = r.uniform(-1, 1, size=(3, len(nodes), 2, 2))
        blk.set_source(nodes, 0.5 * (sv + np.swapaxes(sv, -1, -2)))
        blk.set_field(_lib.FIELD_U, r.uniform(-1, 1, blk.field_shape(_lib.FIELD_U)))
        s0 = r.uniform(-1, 1, blk.field_shape(_lib.FIELD_S))
        blk.set_field(_lib.FIELD_S, 0.5 * (s0 + np.swapaxes(s0, -1, -2)))
        blk.step(3)
        res[path] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    assert np.isfinite(res["generic"][0]).all()
    assert rel_err(res["tile"][0], res["generic"][0]) < TOL
    assert rel_err(res["tile"][1], res["generic"][1]) < TOL


@pytest.mark.parametrize("degree,n,diagonal", [(1, (21, 5), "left"), (2, (9, 6), "right"), (3, (18, 4), "left"), (4, (5, 7), "left"),
                                               (2, (19, 4), "quadrilateral"), (4, (6, 5), "quadrilateral")])
def test_tile_sponge_constant_on_a_cell_needs_no_matrix(gpu, monkeypatch, degree, n, diagonal):
    """A sigma that is one value on all nodes of a cell (the reference's sponges: `x <= 20 ? 1000 : 0` interpolated into
    DG4, explosive_source_lf4.py:42-45) makes the cell's sponge matrix sigma I: the tile kernels then take sigma u at the
    node itself (StageArgs::sponge_sigma).  Cells of all three kinds in one block - none, constant, varying - and
    neighbouring each other inside one 16-cell item: against the generic kernel (every sponge cell through its matrix)
    and against the oracle."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    h = (0.9 / n[0], 1.1 / n[1])
    dt = 0.05 * min(h) / degree ** 2
    r = np.random.default_rng(11)
    m = oracle_mesh(2, n, (0.9, 1.1), diagonal)
    nq = m.node_coords(4).shape[1]
    kind = r.integers(0, 3, size=m.ncells)                     # 0 none, 1 constant, 2 varying
    sigma = np.zeros((m.ncells, nq))
    sigma[kind == 1] = r.uniform(2.0, 30.0, size=((kind == 1).sum(), 1))
    sigma[kind == 2] = r.uniform(0.0, 30.0, size=((kind == 2).sum(), nq))
    u0 = r.uniform(-1, 1, (m.ncells,) + (HipBlock(2, degree, n, h, (0.0, 0.0), diagonal).field_shape(_lib.FIELD_U)[1:]))
    s0 = r.uniform(-1, 1, u0.shape + (2,))
    s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
    res = {}
    for path in ("generic", "tile"):
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        blk = HipBlock(2, degree, n, h, (0.0, 0.0), diagonal)
        blk.set_params(1.0, dt, 0.6, 0.3)
        blk.set_absorption(sigma, 4)
        blk.set_field(_lib.FIELD_U, u0)
        blk.set_field(_lib.FIELD_S, s0)
        blk.step(3)
        res[path] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    orc = OracleLF4(m, degree)
    orc.dt, orc.l, orc.mu, orc.density = dt, 0.6, 0.3, 1.0
    orc.E.set_absorption(sigma, 4)
    orc.u0, orc.s0 = u0.copy(), s0.copy()
    for k in range(3):
        orc.step((k + 1) * dt)
    tol = 10 * (5 * TOL if (diagonal == "quadrilateral" and degree == 4) else TOL)
    assert rel_err(res["tile"][0], res["generic"][0]) < tol
    assert rel_err(res["tile"][1], res["generic"][1]) < tol
    assert rel_err(res["tile"][0], orc.u1) < tol
    assert rel_err(res["tile"][1], orc.s1) < tol


@pytest.mark.parametrize("degree,n,grid", [
    (1, (20, 6), (2, 2)),
    (2, (33, 4), (2, 1)),
    (3, (10, 9), (1, 3)),
    (4, (17, 6), (3, 2)),
])
@pytest.mark.parametrize("pipelined", [True, False])
def test_tile_multiblock_equals_single_block(gpu, tile, degree, n, grid, pipelined):
    """Blocks with neighbours read packed remote traces (T.n records): bitwise the single-block result, with a
    sponge and a scattered source on top."""
    from tests.test_harness_gpu import _multiblock_case
    _multiblock_case(2, degree, n, grid, pipelined, extras=True)


def test_tile_source_fused_equals_source_launch(gpu, monkeypatch):
    """The source added inside the G stage kernels (default) against the same run with the source as a launch of
    its own after every G stage (SEIGEN_HIP_SOURCE_LAUNCH=1): same nodes, same values, rounding-level agreement;
    a node listed twice keeps the separate launch and adds both entries."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    monkeypatch.setenv("SEIGEN_HIP_PATH", "tile")
    n, degree = (23, 7), 3
    h = (1.0 / n[0], 1.0 / n[1])
    res = {}
    for mode in ("fused", "launch", "duplicate"):
        if mode == "launch":
            monkeypatch.setenv("SEIGEN_HIP_SOURCE_LAUNCH", "1")
        else:
            monkeypatch.delenv("SEIGEN_HIP_SOURCE_LAUNCH", raising=False)
        blk = HipBlock(2, degree, n, h, (0.0, 0.0), "left")
        r = np.random.default_rng(9)
        blk.set_params(1.0, 0.05 * min(h) / degree ** 2, 0.5, 0.25)
        nodes = np.unique(r.integers(0, blk.ncells * blk.nd, size=60))
        sv = r.uniform(-1, 1, size=(4, len(nodes), 2, 2))
        sv = 0.5 * (sv + np.swapaxes(sv, -1, -2))
        if mode == "duplicate":      # every node twice with half the value: the same source
            nodes = np.concatenate([nodes, nodes])
            sv = np.concatenate([0.5 * sv, 0.5 * sv], axis=1)
        blk.set_source(nodes, sv)
        blk.set_field(_lib.FIELD_U, r.uniform(-1, 1, blk.field_shape(_lib.FIELD_U)))
        blk.step(4)
        res[mode] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    for mode in ("launch", "duplicate"):
        assert rel_err(res[mode][0], res["fused"][0]) < 1e-13
        assert rel_err(res[mode][1], res["fused"][1]) < 1e-13
    assert np.abs(res["fused"][1]).max() > 0.1
