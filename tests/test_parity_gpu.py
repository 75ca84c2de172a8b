"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs.  FP64 everywhere; tolerance = 1e-11 relative to the largest
entry of the expected field per operator application (different summation
order only), looser bounds stated where many steps accumulate."""
import os

import numpy as np
import pytest

from oracle.forms import ElasticOperators
from oracle.lf4 import OracleLF4
from tests.util import oracle_mesh, rel_err, seeded

pytestmark = pytest.mark.gpu

TOL = 1e-11


def tol_of(degree, diagonal):
    """1e-11, except DQ_4 (5e-11): the oracle inverts each cell's 25 x 25 mass matrix of the equispaced tensor basis
    in double precision; the library builds the element from long-double interval operators."""
    return 5 * TOL if (diagonal == "quadrilateral" and degree == 4) else TOL

CASES = [
    # dim, degree, n, L, diagonal
    (1, 1, (7,), (2.0,), "left"),
    (1, 3, (5,), (1.0,), "left"),
    (2, 1, (4, 4), (1.0, 1.0), "left"),
    (2, 2, (4, 3), (1.0, 1.5), "left"),
    (2, 2, (3, 4), (2.0, 1.0), "right"),
    (2, 3, (3, 3), (1.0, 1.0), "left"),
    (2, 4, (4, 4), (1.0, 1.0), "left"),
    (3, 1, (2, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 2, (2, 3, 2), (1.0, 1.5, 0.5), "left"),
    (3, 3, (2, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 4, (2, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 4, (3, 1, 2), (1.0, 1.0, 1.0), "left"),
    # tall ragged blocks: the chunked XCD order of the F stages (from 16 layers up), groups straddling rows and layers
    (3, 3, (5, 3, 17), (1.0, 0.6, 3.4), "left"),
    (3, 4, (3, 2, 19), (0.6, 0.4, 3.8), "left"),
    # quadrilateral cells, tensor-product element DQ_k (sg_config::diagonal = 2)
    (2, 1, (4, 3), (1.0, 1.5), "quadrilateral"),
    (2, 2, (5, 4), (2.0, 1.0), "quadrilateral"),
    (2, 3, (3, 3), (1.0, 1.0), "quadrilateral"),
    (2, 4, (4, 5), (1.0, 1.0), "quadrilateral"),
    # hexahedral cells, DQ_1..4 (dim 3 with sg_config::diagonal = 2)
    (3, 1, (3, 2, 4), (1.0, 1.5, 0.8), "quadrilateral"),
    (3, 2, (2, 3, 2), (1.0, 0.9, 0.5), "quadrilateral"),
    (3, 2, (5, 1, 3), (1.0, 0.3, 0.9), "quadrilateral"),
    (3, 3, (2, 2, 3), (1.0, 0.8, 0.9), "quadrilateral"),
    (3, 4, (2, 1, 2), (0.6, 0.4, 0.5), "quadrilateral"),
]


def make_block(dim, degree, n, L, diagonal):
    from seigen_amd.backend import HipBlock
    h = [L[a] / n[a] for a in range(dim)]
    return HipBlock(dim, degree, n, h, [0.0] * dim, diagonal)


@pytest.mark.parametrize("dim,degree,n,L,diagonal", CASES)
def test_node_coords_match_oracle(gpu, dim, degree, n, L, diagonal):
    blk = make_block(dim, degree, n, L, diagonal)
    m = oracle_mesh(dim, n, L, diagonal)
    assert blk.ncells == m.ncells
    np.testing.assert_allclose(blk.node_coords(), m.node_coords(degree), rtol=0, atol=1e-13)


@pytest.mark.parametrize("dim,degree,n,L,diagonal", CASES)
def test_apply_F_and_G(gpu, dim, degree, n, L, diagonal):
    from seigen_amd import _lib
    blk = make_block(dim, degree, n, L, diagonal)
    m = oracle_mesh(dim, n, L, diagonal)
    E = ElasticOperators(m, degree)
    T = seeded(blk.field_shape(_lib.FIELD_S), 0)
    u = seeded(blk.field_shape(_lib.FIELD_U), 1)
    lam, mu = 0.7, 0.3
    blk.set_params(1.0, 0.01, lam, mu)
    blk.set_field(_lib.FIELD_S, T)
    blk.set_field(_lib.FIELD_U, u)
    blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
    got = blk.get_field(_lib.FIELD_UH)
    exp = E.apply_F(T, u)
    assert rel_err(got, exp) < tol_of(degree, diagonal)
    blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
    got = blk.get_field(_lib.FIELD_SH)
    exp = E.apply_G(u, lam, mu)
    assert rel_err(got, exp) < tol_of(degree, diagonal)


@pytest.mark.parametrize("dim,degree,n,L,diagonal", CASES)
def test_full_steps(gpu, dim, degree, n, L, diagonal):
    """Three whole LF4 steps (six fused launches each) against the un-fused oracle."""
    from seigen_amd import _lib
    blk = make_block(dim, degree, n, L, diagonal)
    m = oracle_mesh(dim, n, L, diagonal)
    orc = OracleLF4(m, degree)
    hmin = min(L[a] / n[a] for a in range(dim))
    orc.dt = 0.05 * hmin / degree ** 2
    orc.l, orc.mu, orc.density = 0.5, 0.25, 1.0
    orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 2)
    orc.s0 = seeded(blk.field_shape(_lib.FIELD_S), 3)
    blk.set_params(orc.density, orc.dt, orc.l, orc.mu)
    blk.set_field(_lib.FIELD_U, orc.u0)
    blk.set_field(_lib.FIELD_S, orc.s0)
    blk.step(3)
    for k in range(3):
        orc.step((k + 1) * orc.dt)
    tol = 10 * tol_of(degree, diagonal)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < tol
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < tol
    # intermediate fields left behind by the last step
    assert rel_err(blk.get_field(_lib.FIELD_UH), orc.dt * orc.u1 + orc.dt ** 3 / 24.0 * orc.last["utemp"]) < tol
    assert rel_err(blk.get_field(_lib.FIELD_SH), orc.last["sh1"]) < tol


def test_density_quirk(gpu):
    """Explicit mode keeps only rhs(form_u1): u1 = rho*u0 + dt*uh1 + ... (elastic.py:341-345)."""
    from seigen_amd import _lib
    dim, degree, n, L = 2, 2, (3, 3), (1.0, 1.0)
    blk = make_block(dim, degree, n, L, "left")
    orc = OracleLF4(oracle_mesh(dim, n, L), degree)
    orc.dt, orc.l, orc.mu, orc.density = 0.002, 0.5, 0.25, 1.7
    orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 4)
    orc.s0 = seeded(blk.field_shape(_lib.FIELD_S), 5)
    blk.set_params(orc.density, orc.dt, orc.l, orc.mu)
    blk.set_field(_lib.FIELD_U, orc.u0)
    blk.set_field(_lib.FIELD_S, orc.s0)
    blk.step(2)
    orc.step(orc.dt)
    orc.step(2 * orc.dt)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 10 * TOL
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 10 * TOL


def test_against_committed_golden_vectors(gpu):
    """HIP path vs tests/golden/stage_vectors.npz (written by tests/golden/make_golden.py from
    the oracle): un-fused F and G, and the state after 1 and 10 whole LF4 steps."""
    import os
    from seigen_amd import _lib
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage_vectors.npz"))
    ncases = len([k for k in d.files if k.endswith("_meta")])
    assert ncases >= 7
    diag = {3: "right"}
    for ci in range(ncases):
        key = "c%d" % ci
        dim, P = int(d[key + "_meta"][0]), int(d[key + "_meta"][1])
        n = tuple(int(x) for x in d[key + "_meta"][2:])
        L = tuple(d[key + "_L"])
        blk = make_block(dim, P, n, L, diag.get(ci, "left"))
        blk.set_params(1.0, float(d[key + "_dt"]), 0.7, 0.3)
        blk.set_field(_lib.FIELD_S, d[key + "_T"])
        blk.set_field(_lib.FIELD_U, d[key + "_u"])
        blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
        assert rel_err(blk.get_field(_lib.FIELD_UH), d[key + "_F"]) < TOL
        blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
        assert rel_err(blk.get_field(_lib.FIELD_SH), d[key + "_G"]) < TOL
        blk.set_params(1.0, float(d[key + "_dt"]), 0.5, 0.25)
        blk.step(1)
        assert rel_err(blk.get_field(_lib.FIELD_U), d[key + "_u_step1"]) < 10 * TOL
        assert rel_err(blk.get_field(_lib.FIELD_S), d[key + "_s_step1"]) < 10 * TOL
        blk.step(9)
        assert rel_err(blk.get_field(_lib.FIELD_U), d[key + "_u_step10"]) < 100 * TOL
        assert rel_err(blk.get_field(_lib.FIELD_S), d[key + "_s_step10"]) < 100 * TOL


def test_symmetric_stress_mode_and_fallback(gpu):
    """3-D P3/P4 run in symmetric-stress mode (only the i <= j lines of stress fields are touched).
    A non-symmetric upload must leave that mode: F then sees the antisymmetric part exactly as the
    reference's full TensorFunctionSpace does (elastic.py:81)."""
    from seigen_amd import _lib
    dim, degree, n, L = 3, 4, (2, 2, 3), (1.0, 1.0, 1.0)
    m = oracle_mesh(dim, n, L)
    for symmetric in (True, False):
        blk = make_block(dim, degree, n, L, "left")
        orc = OracleLF4(m, degree)
        orc.dt, orc.l, orc.mu, orc.density = 1e-3, 0.5, 0.25, 1.0
        orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 21)
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 22)
        if symmetric:
            s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        orc.s0 = s0.copy()
        blk.set_params(1.0, orc.dt, orc.l, orc.mu)
        blk.set_field(_lib.FIELD_U, orc.u0)
        blk.set_field(_lib.FIELD_S, s0)
        # what comes back is what went in, mirrored lines included
        np.testing.assert_array_equal(blk.get_field(_lib.FIELD_S), s0)
        blk.step(2)
        orc.step(orc.dt)
        orc.step(2 * orc.dt)
        assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 10 * TOL
        assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 10 * TOL
        assert rel_err(blk.get_field(_lib.FIELD_SH), orc.last["sh1"]) < 10 * TOL
        if symmetric:
            # switch mid-run: upload a non-symmetric stress now and keep stepping
            s_now = blk.get_field(_lib.FIELD_S)
            s_now[..., 0, 1] += 0.125
            blk.set_field(_lib.FIELD_S, s_now)
            orc.s0 = s_now.copy()
            orc.step(3 * orc.dt)
            blk.step(1)
            assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 10 * TOL
            assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 10 * TOL


def test_symmetric_mode_lane_path_2d(gpu, monkeypatch):
    """The lane-per-cell kernels (forced here on a small mesh) in symmetric-stress mode, and their
    exact fallback for a non-symmetric stress."""
    from seigen_amd import _lib
    monkeypatch.setenv("SEIGEN_HIP_PATH", "lane")
    dim, degree, n, L = 2, 3, (5, 4), (1.0, 1.25)
    m = oracle_mesh(dim, n, L)
    for symmetric in (True, False):
        blk = make_block(dim, degree, n, L, "left")
        orc = OracleLF4(m, degree)
        orc.dt, orc.l, orc.mu, orc.density = 1e-3, 0.5, 0.25, 1.0
        orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 31)
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 32)
        if symmetric:
            s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        orc.s0 = s0.copy()
        blk.set_params(1.0, orc.dt, orc.l, orc.mu)
        blk.set_field(_lib.FIELD_U, orc.u0)
        blk.set_field(_lib.FIELD_S, s0)
        np.testing.assert_array_equal(blk.get_field(_lib.FIELD_S), s0)
        blk.step(3)
        for k in range(3):
            orc.step((k + 1) * orc.dt)
        assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 10 * TOL
        assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 10 * TOL


@pytest.mark.parametrize("path", ["generic", "lane"])
@pytest.mark.parametrize("dim,degree,n,L,diagonal", [
    (1, 2, (9,), (1.0,), "left"),
    (2, 1, (5, 3), (1.0, 1.0), "left"),
    (2, 4, (3, 4), (1.5, 1.0), "right"),
    (3, 1, (3, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 2, (2, 3, 2), (1.0, 1.5, 0.5), "left"),
])
def test_every_kernel_family(gpu, monkeypatch, path, dim, degree, n, L, diagonal):
    """The size-based choice between the generic and the lane-per-cell kernels must not matter:
    both families, forced through SEIGEN_HIP_PATH, against the oracle (3-D P3/P4 have the MFMA
    kernels, covered by the default runs above)."""
    from seigen_amd import _lib
    monkeypatch.setenv("SEIGEN_HIP_PATH", path)
    blk = make_block(dim, degree, n, L, diagonal)
    m = oracle_mesh(dim, n, L, diagonal)
    orc = OracleLF4(m, degree)
    orc.dt = 0.05 * min(L[a] / n[a] for a in range(dim)) / degree ** 2
    orc.l, orc.mu, orc.density = 0.5, 0.25, 1.0
    orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 41)
    orc.s0 = seeded(blk.field_shape(_lib.FIELD_S), 42)
    blk.set_params(orc.density, orc.dt, orc.l, orc.mu)
    blk.set_field(_lib.FIELD_U, orc.u0)
    blk.set_field(_lib.FIELD_S, orc.s0)
    blk.step(2)
    orc.step(orc.dt)
    orc.step(2 * orc.dt)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 10 * TOL
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 10 * TOL


@pytest.mark.parametrize("path", [None, "generic", "lane"])
@pytest.mark.parametrize("dim,degree,n,L,diagonal", [
    (2, 1, (5, 3), (1.0, 1.0), "left"),
    (2, 3, (4, 4), (1.0, 1.0), "left"),
    (2, 2, (4, 3), (1.0, 1.5), "quadrilateral"),
    (3, 1, (3, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 4, (2, 2, 2), (1.0, 1.0, 1.0), "left"),
    (3, 2, (2, 3, 2), (1.0, 0.9, 0.5), "quadrilateral"),
    (3, 3, (2, 2, 3), (1.0, 0.8, 0.9), "quadrilateral"),
])
def test_stages_utemp_and_s1_read_only_their_operands(gpu, monkeypatch, path, dim, degree, n, L, diagonal):
    """Stage UTEMP writes w = dt u1 + dt^3/24 Minv f(sh1) into UH (include/seigen_hip.h, enum sg_stage) and has no
    self term: whatever UH held before - here NaN - must not reach the result, in any kernel family."""
    from seigen_amd import _lib
    if path:
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
    blk = make_block(dim, degree, n, L, diagonal)
    dt = 0.05 * min(L[a] / n[a] for a in range(dim)) / degree ** 2
    blk.set_params(1.0, dt, 0.5, 0.25)
    u1 = seeded(blk.field_shape(_lib.FIELD_U), 51)
    sh1 = seeded(blk.field_shape(_lib.FIELD_S), 52)
    sh1 = 0.5 * (sh1 + np.swapaxes(sh1, -1, -2))
    results = []
    for fill in (0.0, np.nan):
        blk.set_field(_lib.FIELD_U, u1)
        blk.set_field(_lib.FIELD_SH, sh1)
        blk.set_field(_lib.FIELD_UH, np.full(blk.field_shape(_lib.FIELD_UH), fill))
        blk.run_stage(_lib.STAGE_UTEMP)
        results.append(blk.get_field(_lib.FIELD_UH))
    assert np.isfinite(results[1]).all()
    assert np.array_equal(results[0], results[1])
    # ... and stage S1 = s0 + Minv g(w) reads UH (w) and S only: SH (sh1) is no operand of it any more
    w = results[0]
    s0 = 0.5 * (sh1 + 0.25)
    results = []
    for fill in (0.0, np.nan):
        blk.set_field(_lib.FIELD_UH, w)
        blk.set_field(_lib.FIELD_S, s0)
        blk.set_field(_lib.FIELD_SH, np.full(blk.field_shape(_lib.FIELD_SH), fill))
        blk.run_stage(_lib.STAGE_S1)
        results.append(blk.get_field(_lib.FIELD_S))
    assert np.isfinite(results[1]).all()
    assert np.array_equal(results[0], results[1])
    blk.close()


@pytest.mark.parametrize("degree,n,diagonal,family", [
    (4, (3, 2, 2), "left", "mfma_stage_F"), (3, (5, 2, 3), "left", "mfma_stage_F"), (4, (17, 2, 1), "left", "mfma_stage_F"),
    (2, (5, 3, 2), "left", "lane"), (1, (9, 2, 2), "left", "lane"),                                   # lane-per-cell kernels
    (2, (5, 2, 3), "quadrilateral", "lane"), (1, (7, 3, 2), "quadrilateral", "lane"),               # hexahedra DQ_1 / DQ_2
    (3, (5, 2, 2), "quadrilateral", "hexm_stage"), (4, (3, 2, 1), "quadrilateral", "hexm_stage"),   # hexahedra DQ_3 / DQ_4
])
def test_3d_sponge_constant_and_matrix_cells(gpu, monkeypatch, degree, n, diagonal, family):
    """3-D kernel families: a sigma that is one value on all nodes of a cell is applied as sigma u at the node itself
    (StageArgs::sponge_sigma), a varying one through B_e u computed by a launch of its own before the stage
    (launch_sponge_pre) - cells of all three kinds (none, constant, varying) side by side inside one 16-cell item, three
    whole steps (the in-place stage U1 reads u_abs = its own output buffer) against the generic kernels (every sponge cell
    through its matrix, in the kernel) and against the oracle.  (The same mix of cells on SPLIT stages - the SECOND region
    reads the pre-pass the FIRST region launched, stages.cpp sponge_pre_key / sponge_pre_regions - is checked bitwise
    against the single block by test_harness_gpu.py::_multiblock_case(extras=True) for every family, tetrahedra and
    hexahedra, and over the native exchange by test_native_exchange_gpu.py's "source" cases.)"""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    L = tuple(0.4 * k for k in n)
    h = [L[a] / n[a] for a in range(3)]
    r = np.random.default_rng(21)
    m = oracle_mesh(3, n, L, diagonal)
    nq = m.node_coords(4).shape[1]
    kind = r.integers(0, 4, size=m.ncells)
    sigma = np.zeros((m.ncells, nq))
    sigma[kind == 1] = r.uniform(2.0, 30.0, size=((kind == 1).sum(), 1))
    sigma[kind == 2] = r.uniform(0.0, 30.0, size=((kind == 2).sum(), nq))
    # kind 3 (round 6): a sigma that is AFFINE in x, varying in all three coordinates, a different one per cell - such cells
    # take four coefficients and the element-constant matrices X_k instead of a matrix of their own (sponge_pre_affine_kernel)
    Xq = m.node_coords(4)
    n3 = int((kind == 3).sum())
    grad = r.uniform(-20.0, 20.0, size=(n3, 1, 3))
    sigma[kind == 3] = r.uniform(5.0, 30.0, size=(n3, 1)) + (grad * (Xq[kind == 3] - Xq[kind == 3][:, :1])).sum(axis=-1)
    dt = 0.04 * min(h) / degree ** 2
    res = {}
    for path in ("generic", ""):
        if path:
            monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        elif family == "lane":
            monkeypatch.setenv("SEIGEN_HIP_PATH", "lane")
        else:
            monkeypatch.delenv("SEIGEN_HIP_PATH", raising=False)
        blk = HipBlock(3, degree, n, h, [0.0] * 3, diagonal)
        if not path:
            name = blk.stage_kernel_name(_lib.STAGE_UH1)
            assert (("lane_stage" in name or "hex_stage" in name) if family == "lane" else family in name), name
        u0 = seeded(blk.field_shape(_lib.FIELD_U), 61)
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 62)
        s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
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
    tol = 10 * tol_of(degree, diagonal) * (2 if diagonal == "quadrilateral" else 1)
    assert rel_err(res[""][0], res["generic"][0]) < tol
    assert rel_err(res[""][1], res["generic"][1]) < tol
    assert rel_err(res[""][0], orc.u1) < tol
    assert rel_err(res[""][1], orc.s1) < tol


@pytest.mark.parametrize("degree,n,diagonal,path", [
    (4, (17, 3, 2), "left", ""), (3, (6, 3, 4), "left", ""), (2, (5, 3, 4), "left", "lane"), (1, (9, 2, 3), "left", "lane"),
    (3, (5, 3, 2), "quadrilateral", ""), (4, (3, 2, 2), "quadrilateral", ""), (2, (5, 3, 2), "quadrilateral", "lane"),
])
@pytest.mark.parametrize("sigma_degree", [4, 1])
def test_3d_affine_sigma_ramp(gpu, monkeypatch, degree, n, diagonal, path, sigma_degree):
    """A sponge that is a linear ramp in all three coordinates (sigma in DG4 as the reference's scripts declare it,
    elastic.py:136-141, and in DG1): every cell's sigma is affine in its reference coordinates, so the 3-D families apply
    B_e = s_0 I + sum_k s_k X_k from four coefficients per cell (kernels.hip sponge_pre_affine_kernel) instead of reading an
    nd x nd matrix per cell.  Three whole steps against the SAME library with the affine path switched off (every cell through
    its matrix, SEIGEN_HIP_SPONGE_AFFINE=0) and against the oracle's absorption_matrix form."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    L = tuple(0.4 * k for k in n)
    h = [L[a] / n[a] for a in range(3)]
    m = oracle_mesh(3, n, L, diagonal)
    Xq = m.node_coords(sigma_degree)
    sigma = 4.0 + 11.0 * Xq[..., 0] + 7.0 * Xq[..., 1] + 23.0 * Xq[..., 2]
    dt = 0.04 * min(h) / degree ** 2
    if path:
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
    res = {}
    for affine in ("1", "0"):
        monkeypatch.setenv("SEIGEN_HIP_SPONGE_AFFINE", affine)
        blk = HipBlock(3, degree, n, h, [0.0] * 3, diagonal)
        u0 = seeded(blk.field_shape(_lib.FIELD_U), 71)
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 72)
        s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        blk.set_params(1.0, dt, 0.6, 0.3)
        blk.set_absorption(sigma, sigma_degree)
        blk.set_field(_lib.FIELD_U, u0)
        blk.set_field(_lib.FIELD_S, s0)
        blk.step(3)
        res[affine] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    orc = OracleLF4(m, degree)
    orc.dt, orc.l, orc.mu, orc.density = dt, 0.6, 0.3, 1.0
    orc.E.set_absorption(sigma, sigma_degree)
    orc.u0, orc.s0 = u0.copy(), s0.copy()
    for k in range(3):
        orc.step((k + 1) * dt)
    tol = 10 * tol_of(degree, diagonal) * (2 if diagonal == "quadrilateral" else 1)
    assert np.abs(res["1"][0] - u0).max() > 1e-6
    for a, b in ((res["1"][0], res["0"][0]), (res["1"][1], res["0"][1]), (res["1"][0], orc.u1), (res["1"][1], orc.s1)):
        assert rel_err(a, b) < tol
    # the sponge really acted, and differently from a cell-constant one of the same mean
    assert rel_err(res["1"][0], u0) > 1e-4


def test_error_behaviour(gpu):
    """Every entry point returns a negative code with a message instead of launching on bad input
    (the reference raises Python exceptions: seigen/elastic.py:64, :234-242)."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    blk = HipBlock(3, 4, (4, 2, 2), (0.25, 0.5, 0.5), (0.0, 0.0, 0.0))
    with pytest.raises(_lib.SeigenHipError, match="sg_set_params"):
        blk.step(1)                                           # parameters not set
    blk.set_params(1.0, 1e-3, 0.5, 0.25)
    with pytest.raises(ValueError):
        blk.set_field(_lib.FIELD_U, np.zeros((3, 35, 3)))     # wrong size, caught by the host layer
    small = np.zeros((3, 35, 3))
    assert blk.lib.sg_set_field(blk.h, _lib.FIELD_U, small.ctypes.data, small.nbytes) < 0    # and by the library
    assert b"" != blk.lib.sg_last_error(blk.h)
    assert blk.lib.sg_set_field_range(blk.h, _lib.FIELD_U, 90, 10, small.ctypes.data, small.nbytes) < 0   # past the end
    with pytest.raises(_lib.SeigenHipError):
        blk.run_stage(7)                                      # unknown stage
    with pytest.raises(_lib.SeigenHipError):
        blk.run_stage(0, 9)                                   # unknown region
    with pytest.raises(_lib.SeigenHipError):
        blk.apply_F(_lib.FIELD_U, _lib.FIELD_U, _lib.FIELD_UH)   # s_in must be a stress field
    with pytest.raises(_lib.SeigenHipError):
        blk.set_source([10 ** 9], np.zeros((1, 1, 3, 3)))     # node index out of range
    blk.step(1)                                               # still usable afterwards
    assert np.isfinite(blk.get_field(_lib.FIELD_U)).all()
    # a block with halo neighbours refuses to run without attached halo buffers, and sg_step on it
    nb = HipBlock(3, 4, (4, 2, 2), (0.25, 0.5, 0.5), (0.0, 0.0, 0.0), "left", 0b100000)
    nb.set_params(1.0, 1e-3, 0.5, 0.25)
    with pytest.raises(_lib.SeigenHipError, match="neighbours"):
        nb.step(1)
    with pytest.raises(_lib.SeigenHipError, match="halo"):
        nb.run_stage(0, _lib.REGION_FIRST)
    with pytest.raises(_lib.SeigenHipError):
        HipBlock(3, 7, (2, 2, 2), (0.5, 0.5, 0.5), (0.0, 0.0, 0.0))   # unsupported degree


@pytest.mark.parametrize("seed", range(int(os.environ.get("SEIGEN_TEST_RANDOM_CASES", "10"))))
def test_kernel_families_agree_on_random_blocks(gpu, monkeypatch, seed):
    """Differential test: random dimension, degree, (ragged) block size, cell size, diagonal, sponge
    and per-cell material; the kernel families that support the case must agree after two steps."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.integers(2, 4))
    degree = int(rng.integers(1, 5))
    n = tuple(int(x) for x in (rng.integers(1, 20, size=dim) if dim == 3 else rng.integers(1, 40, size=dim)))
    h = tuple(float(x) for x in rng.uniform(0.2, 1.5, size=dim))
    diagonal = ("left", "right", "quadrilateral")[int(rng.integers(0, 3))] if dim == 2 else "left"
    quad = diagonal == "quadrilateral"
    dt = 0.05 * min(h) / degree ** 2
    results = {}
    for path in ("generic", "lane", "mfma", "tile"):
        if path == "mfma" and dim != 3:
            continue
        if path == "tile" and dim != 2:
            continue
        if path == "lane" and ((dim == 3 and degree > 2) or quad):
            continue
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
        blk = HipBlock(dim, degree, n, h, (0.0,) * dim, diagonal)
        r2 = np.random.default_rng(2000 + seed)
        lam = r2.uniform(0.4, 0.9, blk.ncells) if seed % 2 else 0.5
        mu = r2.uniform(0.2, 0.5, blk.ncells) if seed % 2 else 0.25
        blk.set_params(1.0 if seed % 5 else 1.3, dt, lam, mu)
        if seed % 5 in (1, 2):                 # per-cell density, either convention
            blk.set_density(r2.uniform(0.7, 1.6, blk.ncells), physical=(seed % 5 == 2))
        if seed % 3 == 0:
            nq = 25 if quad else {2: 15, 3: 35}[dim]
            blk.set_absorption(np.where(r2.uniform(size=(blk.ncells, nq)) > 0.7, 5.0, 0.0), 4)
        u0 = r2.uniform(-1, 1, blk.field_shape(_lib.FIELD_U))
        s0 = r2.uniform(-1, 1, blk.field_shape(_lib.FIELD_S))
        if seed % 4:                       # mostly symmetric stress (symmetric mode), sometimes not
            s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
        blk.set_field(_lib.FIELD_U, u0)
        blk.set_field(_lib.FIELD_S, s0)
        blk.step(2)
        results[path] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    ref_u, ref_s = results["generic"]
    assert np.isfinite(ref_u).all() and np.isfinite(ref_s).all()
    for path, (u, s) in results.items():
        assert rel_err(u, ref_u) < 1e-11 and rel_err(s, ref_s) < 1e-11, (path, dim, degree, n)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SEIGEN_TEST_ORACLE_FUZZ", "6"))))
def test_random_cases_against_the_oracle(gpu, seed):
    """Oracle fuzz: random dimension, degree, cell type (triangles either diagonal, quadrilaterals, tetrahedra,
    intervals), ragged sizes down to one cube, cell sizes, per-cell or scalar material, density in either convention,
    DG sponge of a random degree, a nodal source table, symmetric or full stress - three whole steps of the default
    kernel family against the numpy oracle."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    rng = np.random.default_rng(7000 + seed)
    dim = int(rng.integers(1, 4))
    degree = int(rng.integers(1, 5))
    n = tuple(int(x) for x in rng.integers(1, {1: 12, 2: 7, 3: 4}[dim], size=dim))
    h = [float(x) for x in rng.uniform(0.3, 1.4, size=dim)]
    diagonal = ("left", "right", "quadrilateral")[int(rng.integers(0, 3))] if dim == 2 else "left"
    L = tuple(h[a] * n[a] for a in range(dim))
    blk = HipBlock(dim, degree, n, h, [0.0] * dim, diagonal)
    m = oracle_mesh(dim, n, L, diagonal)
    orc = OracleLF4(m, degree)
    nc, nd = m.ncells, blk.nd
    assert nc == blk.ncells
    per_cell = bool(rng.integers(0, 2))
    lam = rng.uniform(0.4, 0.9, nc) if per_cell else 0.6
    mu = rng.uniform(0.2, 0.5, nc) if per_cell else 0.3
    rho_mode = int(rng.integers(0, 3))                     # 0: scalar (explicit convention), 1: per cell, 2: per cell, physical
    rho = rng.uniform(0.7, 1.6, nc) if rho_mode else float(rng.uniform(0.7, 1.6))
    orc.dt, orc.l, orc.mu = 0.04 * min(h) / degree ** 2, lam, mu
    orc.density, orc.density_physical = rho, rho_mode == 2
    blk.set_params(rho if not rho_mode else 1.0, orc.dt, lam, mu)
    if rho_mode:
        blk.set_density(rho, physical=(rho_mode == 2))
    if rng.integers(0, 2):
        q = int(rng.integers(1, 7))            # sg_set_absorption takes DG_1 .. DG_6
        nq = m.node_coords(q).shape[1]
        sigma = np.where(rng.uniform(size=(nc, nq)) > 0.5, rng.uniform(1.0, 20.0), 0.0)
        orc.E.set_absorption(sigma, q)
        blk.set_absorption(sigma, q)
    nsteps = 3
    vals = nodes = None
    if rng.integers(0, 2):
        nodes = np.unique(rng.integers(0, nc * nd, size=min(6, nc * nd)))
        vals = rng.uniform(-1, 1, (nsteps, len(nodes), dim, dim))
        if rng.integers(0, 2):
            vals = 0.5 * (vals + np.swapaxes(vals, -1, -2))
        blk.set_source(nodes, vals)
    u0 = seeded(blk.field_shape(_lib.FIELD_U), 100 + seed)
    s0 = seeded(blk.field_shape(_lib.FIELD_S), 200 + seed)
    if rng.integers(0, 3):
        s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
    orc.u0, orc.s0 = u0.copy(), s0.copy()
    blk.set_field(_lib.FIELD_U, u0)
    blk.set_field(_lib.FIELD_S, s0)
    blk.step(nsteps)
    for k in range(nsteps):
        if vals is not None:
            S = np.zeros((nc * nd, dim, dim))
            S[nodes] = vals[k]
            orc.source = lambda t, S=S: S.reshape(nc, nd, dim, dim)
        orc.step((k + 1) * orc.dt)
    tol = 20 * tol_of(degree, diagonal)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < tol, (dim, degree, n, diagonal)
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < tol, (dim, degree, n, diagonal)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SEIGEN_TEST_TRANSFER_FUZZ", "6"))))
def test_random_field_ranges_round_trip(gpu, monkeypatch, seed):
    """Transfer fuzz: random block, kernel family (i.e. device layout: host order, 16- or 64-cell interleaved), FP64 or
    float storage, symmetric or full stress; whole fields and random cell ranges written and read back in any order
    must agree with a host-side copy (float mode: with what float holds)."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    rng = np.random.default_rng(9000 + seed)
    dim = int(rng.integers(1, 4))
    degree = int(rng.integers(1, 5))
    n = tuple(int(x) for x in rng.integers(1, {1: 40, 2: 24, 3: 7}[dim], size=dim))
    diagonal = ("left", "right", "quadrilateral")[int(rng.integers(0, 3))] if dim == 2 else "left"
    path = ("", "generic", "lane")[int(rng.integers(0, 3))]
    quad = diagonal == "quadrilateral"
    if path == "lane" and (quad or (dim == 3 and degree > 2)):
        path = ""
    if path:
        monkeypatch.setenv("SEIGEN_HIP_PATH", path)
    f32_ok = not path and ((dim == 3 and degree >= 2) or dim == 2)
    dtype = "f32" if (f32_ok and rng.integers(0, 2)) else "f64"
    blk = HipBlock(dim, degree, n, [0.5] * dim, [0.0] * dim, diagonal, dtype=dtype)
    hold = (lambda a: a.astype(np.float32).astype(np.float64)) if dtype == "f32" else (lambda a: a)
    host = {}
    for field in (_lib.FIELD_U, _lib.FIELD_S, _lib.FIELD_UH, _lib.FIELD_SH):
        host[field] = np.zeros(blk.field_shape(field))
    order = [int(f) for f in rng.permutation(4)]
    for it in range(10):
        field = order[it % 4]
        shape = blk.field_shape(field)
        if rng.integers(0, 3) == 0:
            a = rng.uniform(-1, 1, shape)
            if field in (_lib.FIELD_S, _lib.FIELD_SH) and rng.integers(0, 2):
                a = 0.5 * (a + np.swapaxes(a, -1, -2))
            blk.set_field(field, a)
            host[field] = hold(a)
        else:
            c0 = int(rng.integers(0, blk.ncells))
            nc = int(rng.integers(1, blk.ncells - c0 + 1))
            a = rng.uniform(-1, 1, (nc,) + shape[1:])
            if field in (_lib.FIELD_S, _lib.FIELD_SH) and rng.integers(0, 2):
                a = 0.5 * (a + np.swapaxes(a, -1, -2))
            blk.set_field_range(field, c0, a)
            host[field][c0:c0 + nc] = hold(a)
        f2 = int(rng.integers(0, 4))
        if rng.integers(0, 2):
            assert np.array_equal(blk.get_field(f2), host[f2]), (dim, degree, n, diagonal, path, dtype, it)
        else:
            c0 = int(rng.integers(0, blk.ncells))
            nc = int(rng.integers(1, blk.ncells - c0 + 1))
            assert np.array_equal(blk.get_field_range(f2, c0, nc), host[f2][c0:c0 + nc]), (dim, degree, n, diagonal, path, dtype, it)
    blk.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SEIGEN_TEST_SOURCE_FUZZ", "6"))))
def test_random_sources_agree_across_kernel_families(gpu, monkeypatch, seed):
    """Source fuzz: a random node list (possibly with a node listed twice - both entries count -, possibly all in one
    cell), as a table per step, one static slice, or a separable pattern x weights; shorter than, equal to or longer
    than the run; symmetric or not.  The default kernel family (fused source on the 2-D tile path, source launches
    elsewhere) against the generic kernels, and the separable form against its table."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    rng = np.random.default_rng(11000 + seed)
    dim = int(rng.integers(1, 4))
    degree = int(rng.integers(1, 5))
    n = tuple(int(x) for x in rng.integers(1, {1: 30, 2: 20, 3: 6}[dim], size=dim))
    diagonal = ("left", "right", "quadrilateral")[int(rng.integers(0, 3))] if dim == 2 else "left"
    nrun = int(rng.integers(2, 7))
    results = {}
    for variant in ("generic", "default", "default-separable"):
        monkeypatch.setenv("SEIGEN_HIP_PATH", "generic" if variant == "generic" else "")
        if variant != "generic":
            monkeypatch.delenv("SEIGEN_HIP_PATH")
        blk = HipBlock(dim, degree, n, [0.7] * dim, [0.0] * dim, diagonal)
        r2 = np.random.default_rng(12000 + seed)
        nn = blk.ncells * blk.nd
        k = int(r2.integers(1, min(12, nn) + 1))
        nodes = r2.integers(0, nn, size=k) if r2.integers(0, 2) else r2.integers(0, min(nn, blk.nd), size=k)
        if r2.integers(0, 3):
            nodes = np.unique(nodes)                       # otherwise duplicates may stay in
        mode = ("table", "static", "separable")[int(r2.integers(0, 3))]
        nsrc = int(r2.integers(1, nrun + 3))
        pat = r2.uniform(-1, 1, (len(nodes), dim, dim))
        if r2.integers(0, 2):
            pat = 0.5 * (pat + np.swapaxes(pat, -1, -2))
        w = r2.uniform(-2, 2, nsrc)
        blk.set_params(1.0, 0.01, 0.5, 0.25)
        blk.set_field(_lib.FIELD_U, r2.uniform(-1, 1, blk.field_shape(_lib.FIELD_U)))
        if mode == "static":
            blk.set_source(nodes, pat[None], static=True)
        elif mode == "separable" and variant == "default-separable":
            blk.set_source_separable(nodes, pat, w)
        else:
            table = w[:, None, None, None] * pat[None] if mode == "separable" else r2.uniform(-1, 1, (nsrc, len(nodes), dim, dim))
            blk.set_source(nodes, table)
        blk.step(nrun)
        results[variant] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    ref = results["generic"]
    assert np.abs(ref[1]).max() > 0
    for variant in ("default", "default-separable"):
        assert rel_err(results[variant][0], ref[0]) < 1e-11 and rel_err(results[variant][1], ref[1]) < 1e-11, (variant, dim, degree, n, diagonal)
    # one handle, same calls: bitwise; the separable form against its table of products: bitwise too, unless a node is
    # listed twice (then the pattern entries are summed before the weight is applied, the table entries after)
    if len(np.unique(nodes)) == len(nodes) or mode != "separable":
        assert np.array_equal(results["default"][1], results["default-separable"][1])
    else:
        assert rel_err(results["default-separable"][1], results["default"][1]) < 1e-13
