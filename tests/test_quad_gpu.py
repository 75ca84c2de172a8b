"""Quadrilateral meshes (tensor-product element DQ_k) on the HIP path - SURVEY 8(f) rank 4.

``ElasticLF4.create(mesh, family, degree, dimension)`` (seigen/elastic.py:27-64) is family-agnostic and builds its
spaces with ``FunctionSpace(mesh, family, degree)`` (:81-82); on ``UnitSquareMesh(N, N, quadrilateral=True)`` that is
[upstream] the tensor product of two interval DG elements.  The reference's tests never use such a mesh, so there is
nothing reference-held to pin: parity is against the oracle's quadrature assembly of the same forms on the same
cells (oracle/refelem.py el_*, oracle/mesh.py kind "tensor"), plus convergence to the analytic eigenmode.
Stage-level parity and whole steps on quadrilaterals are in tests/test_parity_gpu.py (CASES, "quadrilateral")."""
import math

import numpy as np
import pytest

from oracle import mesh as omesh
from oracle.forms import ElasticOperators
from oracle.harness import Eigenmode2D
from oracle.lf4 import OracleLF4
from tests.util import rel_err, seeded

pytestmark = pytest.mark.gpu


def _quiet():
    import seigen_amd
    import seigen_amd.helpers as helpers
    import seigen_amd.harness.eigenmode as he
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None
    he.log = lambda s: None


@pytest.mark.parametrize("P,N", [(1, 8), (2, 8), (3, 4), (4, 4)])
def test_eigenmode_on_quadrilaterals_matches_oracle(gpu, P, N):
    """tests/eigenmode/eigenmode_2d.py on UnitSquareMesh(N, N, quadrilateral=True): the error functional of
    :49-63 through the harness equals the oracle's to 1e-9 (north star: 1e-6)."""
    _quiet()
    from seigen_amd.harness.eigenmode import Eigenmode2DLF4
    dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
    em = Eigenmode2DLF4(N, P, dt, solver="explicit", output=False, quadrilateral=True)
    assert em.elastic.U.nd == (P + 1) ** 2 and em.elastic.U.ncells == N * N
    u1, s1 = em.eigenmode2d(T=5.0)
    u_error, s_error = em.eigenmode_error(u1, s1)
    oe = Eigenmode2D(N, P, dt, quadrilateral=True)
    ou, os_ = oe.run(5.0)
    e = oe.errors(ou, os_)
    assert abs(u_error - e["u_error"]) < 1e-9 and abs(s_error - e["s_error"]) < 1e-9, (u_error, s_error, e)
    # the fields themselves after 80-320 steps: different summation order only (1e-9 at P4, whose equispaced 25-node
    # basis is the worst conditioned)
    assert rel_err(u1.dat.data_cells, ou) < 1e-8 and rel_err(s1.dat.data_cells, os_) < 1e-8


def test_eigenmode_sweep_on_quadrilaterals_matches_goldens_and_converges(gpu):
    """The sweep of eigenmode_2d.py:68-84 (P1..4, N = 4, 8, 16) on quadrilateral meshes against the oracle's committed
    error functionals (tests/golden/eigenmode_errors.json "2d_quadrilateral") to 1e-9, and the observed orders."""
    import json
    import os
    _quiet()
    from seigen_amd.harness.eigenmode import Eigenmode2DLF4
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eigenmode_errors.json")))
    rows = {(r["P"], r["N"]): r for r in gold["2d_quadrilateral"]}
    assert len(rows) == 12
    for P, floor in ((1, 1.0), (2, 2.5), (3, 2.7), (4, 4.5)):
        errs = []
        for N in (4, 8, 16):
            g = rows[(P, N)]
            em = Eigenmode2DLF4(N, P, g["dt"], solver="explicit", output=False, quadrilateral=True)
            u1, s1 = em.eigenmode2d(T=5.0)
            errs.append(em.eigenmode_error(u1, s1))
            assert abs(errs[-1][0] - g["u_error"]) < 1e-9 and abs(errs[-1][1] - g["s_error"]) < 1e-9, (P, N, errs[-1], g)
        ou = math.log2(errs[1][0] / errs[2][0])
        os_ = math.log2(errs[1][1] / errs[2][1])
        assert ou > floor and os_ > floor, (P, ou, os_)


@pytest.mark.parametrize("P,path", [(1, None), (2, None), (2, "generic"), (3, None), (4, None), (4, "generic")])
def test_sponge_source_and_material_on_quadrilaterals(gpu, monkeypatch, P, path):
    """The extras of the explosive-source set-up on quadrilateral cells: DG4 sponge (elastic.py:207-208), a nodal
    source table (:217-218) and per-cell lambda / mu, twelve steps against the oracle - on the MFMA tile kernels
    (the default; DQ_4 with two row tiles) and on the generic kernels (forced)."""
    if path:
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    n, L = (6, 5), (3.0, 2.5)
    h = [L[a] / n[a] for a in range(2)]
    blk = HipBlock(2, P, n, h, [0.0, 0.0], "quadrilateral")
    m = omesh.structured(2, n, L, quadrilateral=True)
    orc = OracleLF4(m, P)
    nc = m.ncells
    rng = np.random.default_rng(3)
    lam, mu = rng.uniform(0.4, 0.8, nc), rng.uniform(0.2, 0.4, nc)
    orc.dt, orc.l, orc.mu, orc.density = 0.02 * min(h) / P ** 2, lam, mu, 1.0
    X4 = m.node_coords(4)
    sigma = np.where(X4[..., 0] < 1.0, 30.0 * (1.0 - X4[..., 0]), 0.0)          # DG4 nodal values [nc, 25]
    orc.E.set_absorption(sigma, 4)
    nsteps = 12
    nodes = np.array([2 * (P + 1) ** 2 + 1, 14 * (P + 1) ** 2 + 3, 14 * (P + 1) ** 2 + 4])
    vals = rng.uniform(-1, 1, (nsteps, len(nodes), 2, 2))
    vals = 0.5 * (vals + np.swapaxes(vals, -1, -2))

    def source(k):
        S = np.zeros((nc * (P + 1) ** 2, 2, 2))
        S[nodes] = vals[k]
        return S.reshape(nc, (P + 1) ** 2, 2, 2)

    orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 6)
    s0 = seeded(blk.field_shape(_lib.FIELD_S), 7)
    orc.s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
    blk.set_params(1.0, orc.dt, lam, mu)
    blk.set_absorption(sigma, 4)
    blk.set_source(nodes, vals)
    blk.set_field(_lib.FIELD_U, orc.u0)
    blk.set_field(_lib.FIELD_S, orc.s0)
    blk.step(nsteps)
    for k in range(nsteps):
        orc.source = lambda t, k=k: source(k)
        orc.step((k + 1) * orc.dt)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 1e-10
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 1e-10


@pytest.mark.parametrize("P", [1, 2, 3, 4])
@pytest.mark.parametrize("n", [(37, 23), (16, 1), (5, 40), (129, 3)])
def test_quadrilateral_tile_kernels_agree_with_the_generic_kernels(gpu, monkeypatch, P, n):
    """Ragged blocks (cell groups of 16 straddling rows, one-row and narrow blocks): the MFMA tile kernels against
    the table-driven generic kernels, with sponge, source, per-cell material and per-cell physical density."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    rng = np.random.default_rng(P * 100 + n[0])
    h = [0.7, 1.3]
    res = {}
    for path in ("generic", "tile"):
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        blk = HipBlock(2, P, n, h, [0.0, 0.0], "quadrilateral")
        nc, nd = blk.ncells, blk.nd
        if path == "generic":
            lam, mu, rho = rng.uniform(0.4, 0.8, nc), rng.uniform(0.2, 0.4, nc), rng.uniform(0.8, 1.6, nc)
            sigma = np.where(rng.uniform(size=(nc, 25)) > 0.7, 20.0, 0.0)
            nodes = np.unique(rng.integers(0, nc * nd, size=min(30, nc * nd)))
            vals = rng.uniform(-1, 1, (5, len(nodes), 2, 2))
            vals = 0.5 * (vals + np.swapaxes(vals, -1, -2))
            u0 = seeded(blk.field_shape(_lib.FIELD_U), 1)
            s0 = seeded(blk.field_shape(_lib.FIELD_S), 2)
            s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        blk.set_params(1.0, 0.01 * min(h) / P ** 2, lam, mu)
        blk.set_density(rho, physical=True)
        blk.set_absorption(sigma, 4)
        blk.set_source(nodes, vals)
        blk.set_field(_lib.FIELD_U, u0)
        blk.set_field(_lib.FIELD_S, s0)
        blk.step(5)
        res[path] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    assert rel_err(res["tile"][0], res["generic"][0]) < 1e-12
    assert rel_err(res["tile"][1], res["generic"][1]) < 1e-12


def test_explosive_source_harness_on_quadrilaterals(gpu):
    """tests/explosive_source/explosive_source_lf4.py on RectangleMesh(..., quadrilateral=True), a 100 m x 50 m cut
    at h = 2.5, P2: box-Ricker source (:36-40), DG4 sponge (:43-45), receivers as uy.py:36-43 - traces and final
    fields against the oracle's run of the same set-up."""
    _quiet()
    from oracle.harness import ExplosiveSource
    from seigen_amd.harness.explosive_source import ExplosiveSourceLF4
    import seigen_amd.harness.explosive_source as hx
    hx.log = lambda s: None
    Lx, Ly, h, dt, nsteps = 100.0, 50.0, 2.5, 0.001, 60
    recv = ((45.0, 49.0), (60.0, 49.0), (52.3, 41.7))
    es = ExplosiveSourceLF4()
    es.setup(Lx, Ly, h, degree=2, dt=dt, quadrilateral=True)
    t, tr = es.record_receivers(nsteps * dt * (1 + 1e-9), receivers=recv, every=5)
    orc = ExplosiveSource(Lx, Ly, h, 2, quadrilateral=True)
    orc.elastic.dt = dt
    ot, otr = orc.run(nsteps * dt * (1 + 1e-9), receivers=recv)
    assert len(ot) == nsteps and len(t) == nsteps // 5
    np.testing.assert_allclose(t, ot[4::5], rtol=0, atol=1e-12)
    scale = np.abs(otr).max()
    assert scale > 0 and np.abs(tr - otr[4::5]).max() < 1e-9 * scale
    assert rel_err(es.elastic.u1.dat.data_cells, orc.elastic.u1) < 1e-9
    assert rel_err(es.elastic.s1.dat.data_cells, orc.elastic.s1) < 1e-9


def test_quadrilateral_blocks_equal_the_single_block(gpu):
    """2 x 2 blocks of quadrilateral cells exchanging packed traces = the single block, bit for bit (the halo
    layer of elastic.py:404-436 is direction- and cell-type-agnostic)."""
    from tests.test_harness_gpu import _multiblock_case
    for pipelined in (True, False):
        _multiblock_case(2, 3, (8, 6), (2, 2), pipelined, extras=True, diagonal="quadrilateral")      # tile kernels
        _multiblock_case(2, 4, (6, 6), (2, 2), pipelined, extras=True, diagonal="quadrilateral")      # generic kernels
    _multiblock_case(2, 2, (7, 5), (3, 1), True, diagonal="quadrilateral")
    _multiblock_case(2, 2, (72, 6), (2, 2), True, extras=True, separable=True, diagonal="quadrilateral")   # x sides, wide rows


def test_config2_setup_on_quadrilaterals_full_size_tile_vs_generic(gpu, monkeypatch):
    """Config 2's set-up (512 x 512 squares, P2, DG4 sponge, box-Ricker source) on quadrilateral cells at full size:
    the MFMA tile kernels against the table-driven generic kernels after 20 steps, and the float tile kernels against
    the double ones (against the oracle at this size: tests/test_fullsize_oracle_gpu.py, golden fullsize_c2q.npz)."""
    _quiet()
    import seigen_amd.harness.explosive_source as hx
    hx.log = lambda s: None
    from seigen_amd.harness.explosive_source import ExplosiveSourceLF4
    res = {}
    for name, path, dtype in (("generic", "generic", "f64"), ("tile", "tile", "f64"), ("tile32", "tile", "f32")):
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        es = ExplosiveSourceLF4()
        el = es.setup(Lx=1280.0, Ly=1280.0, h=2.5, degree=2, courant_number=0.05, quadrilateral=True, dtype=dtype)
        # the wavelet of the reference peaks at 0.3 s = 1500 of these steps: shorten its delay so that 20 steps see it
        el.source_expression = el.source_expression.__class__(
            tuple(tuple(c.replace("t - 0.3", "t - 0.003") for c in row) for row in el.source_expression.code), a=159.42, t=0)
        el.source = el.source_expression
        u1, s1 = el.run(20 * el.dt * (1 + 1e-9))
        res[name] = (u1.dat.data_cells.copy(), s1.dat.data_cells.copy())
        assert el.U.ncells == 512 * 512 and el.U.nd == 9
        el.block.close()
    scale_u, scale_s = np.abs(res["generic"][0]).max(), np.abs(res["generic"][1]).max()
    assert scale_u > 0 and scale_s > 0
    assert np.abs(res["tile"][0] - res["generic"][0]).max() < 1e-11 * scale_u
    assert np.abs(res["tile"][1] - res["generic"][1]).max() < 1e-11 * scale_s
    assert np.abs(res["tile32"][0] - res["generic"][0]).max() < 2e-4 * scale_u
    assert np.abs(res["tile32"][1] - res["generic"][1]).max() < 2e-4 * scale_s


def test_one_cell_block_moves_every_field(gpu):
    """A block of a single quadrilateral: the staging buffer of the interleaved layouts is allocated by the first
    transfer (a velocity field here) and must still hold a cell of the stress field (it was sized by the first field:
    sg_set_field of the stress then made no progress)."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    for P in (1, 2, 3):
        blk = HipBlock(2, P, (1, 1), [1.0, 1.0], [0.0, 0.0], "quadrilateral")
        u = seeded(blk.field_shape(_lib.FIELD_U), 1)
        s = seeded(blk.field_shape(_lib.FIELD_S), 2)
        blk.set_field(_lib.FIELD_U, u)
        blk.set_field(_lib.FIELD_S, s)
        assert np.array_equal(blk.get_field(_lib.FIELD_U), u) and np.array_equal(blk.get_field(_lib.FIELD_S), s)
        blk.set_params(1.0, 1e-3, 0.5, 0.25)
        blk.step(2)
        assert np.isfinite(blk.get_field(_lib.FIELD_S)).all()
        blk.close()
