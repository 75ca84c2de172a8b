"""The FP32 second mode (sg_config.dtype = 1; SURVEY 8b `dtype`, 8d "optional second mode, 32 B per
DoF-update"): float storage and arithmetic on the 3-D MFMA path and on the 2-D MFMA tile kernels - triangles P1-P4,
quadrilaterals DQ_1-4 - (v_mfma_f32_16x16x4_f32).
The reference is FP64 throughout (seigen/elastic.py:442), so this mode is checked against the same
FP64 oracle with float tolerances, stated here:

  * one operator application: 2e-5 of the largest entry (35-term dot products of float products,
    operator entries up to about 60);
  * a few LF4 steps: 5e-5;
  * eigenmode error functional (eigenmode_3d.py:42-69) after 40 steps: within 2e-5 of the FP64 value.

FP64 stays the parity and headline mode; nothing here relaxes a FP64 tolerance.
"""
import numpy as np
import pytest

from oracle.forms import ElasticOperators
from oracle.lf4 import OracleLF4
from tests.util import oracle_mesh, rel_err, seeded

pytestmark = pytest.mark.gpu

CASES = [(2, (2, 3, 2), (1.0, 1.5, 0.5)), (3, (2, 2, 2), (1.0, 1.0, 1.0)), (4, (2, 2, 2), (1.0, 1.0, 1.0)),
         (4, (3, 1, 2), (1.0, 1.0, 1.0)), (3, (17, 2, 1), (2.0, 1.0, 1.0))]


def _block(degree, n, L, **kw):
    from seigen_amd.backend import HipBlock
    return HipBlock(3, degree, n, [L[a] / n[a] for a in range(3)], [0.0] * 3, dtype="f32", **kw)


@pytest.mark.parametrize("degree,n,L", CASES)
def test_fp32_operators_against_the_fp64_oracle(gpu, degree, n, L):
    from seigen_amd import _lib
    blk = _block(degree, n, L)
    E = ElasticOperators(oracle_mesh(3, n, L), degree)
    T = seeded(blk.field_shape(_lib.FIELD_S), 0)
    T = 0.5 * (T + np.swapaxes(T, -1, -2))
    u = seeded(blk.field_shape(_lib.FIELD_U), 1)
    blk.set_params(1.0, 0.01, 0.7, 0.3)
    blk.set_field(_lib.FIELD_S, T)
    blk.set_field(_lib.FIELD_U, u)
    # fields come back as what float holds
    assert rel_err(blk.get_field(_lib.FIELD_U), u.astype(np.float32).astype(np.float64)) == 0.0
    blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
    assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(T, u)) < 2e-5
    blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
    assert rel_err(blk.get_field(_lib.FIELD_SH), E.apply_G(u, 0.7, 0.3)) < 2e-5
    # the full-tensor kernels (a non-symmetric stress leaves symmetric storage)
    Ta = seeded(blk.field_shape(_lib.FIELD_S), 5)
    blk.set_field(_lib.FIELD_S, Ta)
    assert not blk.is_sym()
    blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
    assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(Ta, u)) < 2e-5


@pytest.mark.parametrize("degree,n,L", CASES[:4])
def test_fp32_full_steps(gpu, degree, n, L):
    from seigen_amd import _lib
    blk = _block(degree, n, L)
    orc = OracleLF4(oracle_mesh(3, n, L), degree)
    orc.dt = 0.05 * min(L[a] / n[a] for a in range(3)) / degree ** 2
    orc.l, orc.mu, orc.density = 0.5, 0.25, 1.0
    orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 2)
    s0 = seeded(blk.field_shape(_lib.FIELD_S), 3)
    orc.s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
    blk.set_params(1.0, orc.dt, orc.l, orc.mu)
    blk.set_field(_lib.FIELD_U, orc.u0)
    blk.set_field(_lib.FIELD_S, orc.s0)
    blk.step(3)
    for k in range(3):
        orc.step((k + 1) * orc.dt)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 5e-5
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 5e-5


def test_fp32_eigenmode_error_close_to_fp64(gpu):
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.eigenmode as he
    helpers.log = seigen_amd.elastic.log = he.log = lambda s: None
    N, P = 4, 3
    dt = 0.5 * (1.0 / N) / 2 ** (P - 1)
    errs = {}
    for dtype in ("f64", "f32"):
        em = he.Eigenmode3DLF4(N, P, dt, output=False)
        em.elastic = seigen_amd.ElasticLF4.create(em.mesh, "DG", P, dimension=3, solver="explicit", output=False, dtype=dtype)
        em.elastic.density, em.elastic.dt, em.elastic.mu, em.elastic.l = 1.0, dt, 0.25, 0.5
        u1, s1 = em.eigenmode3d(T=5.0)
        errs[dtype] = em.eigenmode_error(u1, s1)
    assert abs(errs["f32"][0] - errs["f64"][0]) < 2e-5 and abs(errs["f32"][1] - errs["f64"][1]) < 2e-5, errs
    assert errs["f64"][0] < 2e-3


@pytest.mark.parametrize("degree,n,grid", [(4, (4, 2, 4), (2, 1, 2)), (3, (9, 9, 9), (3, 3, 3))])
def test_fp32_multiblock_equals_single_block_bitwise(gpu, degree, n, grid):
    from tests.test_harness_gpu import _multiblock_case
    _multiblock_case(3, degree, n, grid, True, extras=True, dtype="f32")


def test_fp32_needs_an_mfma_path(gpu, monkeypatch):
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    with pytest.raises(_lib.SeigenHipError, match="f32"):
        HipBlock(1, 2, (8,), (0.25,), (0.0,), dtype="f32")                                      # 1-D: lane / generic kernels
    monkeypatch.setenv("SEIGEN_HIP_PATH", "generic")
    with pytest.raises(_lib.SeigenHipError, match="f32"):
        HipBlock(2, 2, (4, 4), (0.25, 0.25), (0.0, 0.0), dtype="f32")


CASES_2D = [(1, (9, 7), "left"), (2, (8, 5), "right"), (3, (17, 3), "left"), (4, (6, 6), "left"),
            (1, (7, 9), "quadrilateral"), (2, (8, 5), "quadrilateral"), (3, (17, 3), "quadrilateral"),
            (4, (7, 5), "quadrilateral")]


@pytest.mark.parametrize("degree,n,diagonal", CASES_2D)
def test_fp32_2d_operators_and_steps_against_the_fp64_oracle(gpu, degree, n, diagonal):
    """2-D tile kernels in float: one application of F and G (symmetric and full-tensor storage), and five whole
    steps with sponge, source, per-cell material and density, against the FP64 oracle at float tolerances."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    L = (1.7, 1.1)
    h = [L[a] / n[a] for a in range(2)]
    blk = HipBlock(2, degree, n, h, [0.0, 0.0], diagonal, dtype="f32")
    m = oracle_mesh(2, n, L, diagonal)
    E = ElasticOperators(m, degree)
    T = seeded(blk.field_shape(_lib.FIELD_S), 0)
    Ts = 0.5 * (T + np.swapaxes(T, -1, -2))
    u = seeded(blk.field_shape(_lib.FIELD_U), 1)
    blk.set_params(1.0, 0.01, 0.7, 0.3)
    blk.set_field(_lib.FIELD_S, Ts)
    blk.set_field(_lib.FIELD_U, u)
    assert blk.is_sym()
    assert rel_err(blk.get_field(_lib.FIELD_S), Ts.astype(np.float32).astype(np.float64)) == 0.0
    blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
    assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(Ts, u)) < 2e-5
    blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
    assert rel_err(blk.get_field(_lib.FIELD_SH), E.apply_G(u, 0.7, 0.3)) < 2e-5
    blk.set_field(_lib.FIELD_S, T)
    assert not blk.is_sym()
    blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
    assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(T, u)) < 2e-5
    blk.close()
    # whole steps with every extra
    blk = HipBlock(2, degree, n, h, [0.0, 0.0], diagonal, dtype="f32")
    orc = OracleLF4(m, degree)
    nc, nd = m.ncells, blk.nd
    rng = np.random.default_rng(8)
    lam, mu, rho = rng.uniform(0.4, 0.8, nc), rng.uniform(0.2, 0.4, nc), rng.uniform(0.8, 1.5, nc)
    orc.dt, orc.l, orc.mu = 0.03 * min(h) / degree ** 2, lam, mu
    orc.density, orc.density_physical = rho, True
    nq = 25 if diagonal == "quadrilateral" else 15
    sigma = np.where(rng.uniform(size=(nc, nq)) > 0.6, 5.0, 0.0)
    orc.E.set_absorption(sigma, 4)
    nodes = np.unique(rng.integers(0, nc * nd, size=12))
    vals = rng.uniform(-1, 1, (5, len(nodes), 2, 2))
    vals = 0.5 * (vals + np.swapaxes(vals, -1, -2))

    def source(k):
        S = np.zeros((nc * nd, 2, 2))
        S[nodes] = vals[k]
        return S.reshape(nc, nd, 2, 2)

    orc.u0, orc.s0 = u.copy(), Ts.copy()
    blk.set_params(1.0, orc.dt, lam, mu)
    blk.set_density(rho, physical=True)
    blk.set_absorption(sigma, 4)
    blk.set_source(nodes, vals)
    blk.set_field(_lib.FIELD_U, u)
    blk.set_field(_lib.FIELD_S, Ts)
    blk.step(5)
    for k in range(5):
        orc.source = lambda t, k=k: source(k)
        orc.step((k + 1) * orc.dt)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 5e-5
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 5e-5


@pytest.mark.parametrize("degree,n,grid,diagonal", [(3, (72, 6), (2, 2), "left"), (2, (8, 6), (2, 2), "quadrilateral")])
def test_fp32_2d_multiblock_equals_single_block_bitwise(gpu, degree, n, grid, diagonal):
    from tests.test_harness_gpu import _multiblock_case
    _multiblock_case(2, degree, n, grid, True, extras=True, dtype="f32", diagonal=diagonal)


def test_fp32_2d_eigenmode_error_close_to_fp64(gpu):
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.eigenmode as he
    helpers.log = seigen_amd.elastic.log = he.log = lambda s: None
    N, P = 8, 2
    dt = 0.5 * (1.0 / N) / 2 ** (P - 1)
    for quad in (False, True):
        errs = {}
        for dtype in ("f64", "f32"):
            em = he.Eigenmode2DLF4(N, P, dt, output=False, quadrilateral=quad)
            em.elastic = seigen_amd.ElasticLF4.create(em.mesh, "DG", P, dimension=2, solver="explicit", output=False, dtype=dtype)
            em.elastic.density, em.elastic.dt, em.elastic.mu, em.elastic.l = 1.0, dt, 0.25, 0.5
            u1, s1 = em.eigenmode2d(T=5.0)
            errs[dtype] = em.eigenmode_error(u1, s1)
        assert abs(errs["f32"][0] - errs["f64"][0]) < 2e-5 and abs(errs["f32"][1] - errs["f64"][1]) < 2e-5, (quad, errs)


@pytest.mark.parametrize("degree,n,diagonal", [(4, (17, 3, 2), "left"), (2, (18, 3, 2), "left"), (3, (6, 3, 4), "left")])
def test_affine_sigma_ramp_fp32(gpu, monkeypatch, degree, n, diagonal):
    """The FP32 second mode with a sponge that is a linear ramp in all three coordinates: its affine cells go through
    sponge_pre_affine_kernel<float> (the matrix-pipe form is double only); against the same library with every cell
    through its matrix, both in float: the difference is float round-off."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    L = tuple(0.4 * k for k in n)
    h = [L[a] / n[a] for a in range(3)]
    Xq = oracle_mesh(3, n, L, diagonal).node_coords(4)
    sigma = 4.0 + 11.0 * Xq[..., 0] + 7.0 * Xq[..., 1] + 23.0 * Xq[..., 2]
    dt = 0.04 * min(h) / degree ** 2
    res = {}
    for affine in ("1", "0"):
        monkeypatch.setenv("SEIGEN_HIP_SPONGE_AFFINE", affine)
        blk = HipBlock(3, degree, n, h, [0.0] * 3, diagonal, dtype="f32")
        u0 = seeded(blk.field_shape(_lib.FIELD_U), 81)
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 82)
        s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        blk.set_params(1.0, dt, 0.6, 0.3)
        blk.set_absorption(sigma, 4)
        blk.set_field(_lib.FIELD_U, u0)
        blk.set_field(_lib.FIELD_S, s0)
        blk.step(3)
        res[affine] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    assert np.isfinite(res["1"][0]).all() and rel_err(res["1"][0], u0) > 1e-4
    assert rel_err(res["1"][0], res["0"][0]) < 2e-5 and rel_err(res["1"][1], res["0"][1]) < 2e-5
