"""Hexahedral meshes (tensor-product element DQ_1..4 on cubes) on the HIP path - SURVEY 8(f), the last "next" row.
Degrees 1 and 2 have two kernel families (sum-factorised lane-per-cell kernels for large blocks, the sum-factorised
thread-per-node generic kernel for small ones), degrees 3 and 4 the generic kernel only.

``ElasticLF4.create(mesh, family, degree, dimension)`` (seigen/elastic.py:27-64) builds its spaces with
``FunctionSpace(mesh, family, degree)`` (:81-82); on a hexahedral mesh that is [upstream] the tensor product of three
interval DG elements.  The reference's tests never use such a mesh, so there is nothing reference-held to pin: parity
is against the oracle's quadrature assembly of the same forms on the same cells (oracle/refelem.py el_*,
oracle/mesh.py kind "tensor"), plus convergence to the analytic eigenmode of tests/eigenmode/eigenmode_3d.py.
Stage-level parity and whole steps on hexahedra are in tests/test_parity_gpu.py (CASES, dim 3 "quadrilateral")."""
import math

import numpy as np
import pytest

from oracle import mesh as omesh
from oracle.harness import Eigenmode3D
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


@pytest.mark.parametrize("P,N", [(1, 4), (2, 4), (3, 2), (4, 2)])
def test_eigenmode_on_hexahedra_matches_oracle(gpu, P, N):
    """tests/eigenmode/eigenmode_3d.py on UnitCubeMesh(N, N, N, hexahedral=True): the error functional of :42-69
    through the harness equals the oracle's to 1e-9 (north star: 1e-6), the fields to 1e-9 relative."""
    _quiet()
    from seigen_amd.harness.eigenmode import Eigenmode3DLF4
    dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
    em = Eigenmode3DLF4(N, P, dt, solver="explicit", output=False, hexahedral=True)
    assert em.elastic.U.nd == (P + 1) ** 3 and em.elastic.U.ncells == N ** 3
    u1, s1 = em.eigenmode3d(T=5.0)
    u_error, s_error = em.eigenmode_error(u1, s1)
    oe = Eigenmode3D(N, P, dt, hexahedral=True)
    ou, os_ = oe.run(5.0)
    e = oe.errors(ou, os_)
    assert abs(u_error - e["u_error"]) < 1e-9 and abs(s_error - e["s_error"]) < 1e-9, (u_error, s_error, e)
    assert rel_err(u1.dat.data_cells, ou) < 1e-9 and rel_err(s1.dat.data_cells, os_) < 1e-9


def test_eigenmode_on_hexahedra_converges(gpu):
    """DQ_2 on 4^3, 8^3 and 16^3 cubes at T = 1: the nodal error against the analytic mode falls at better than
    second order (the oracle's own rate on 4^3 -> 8^3: tests/test_hex_oracle.py)."""
    _quiet()
    from seigen_amd.harness.eigenmode import Eigenmode3DLF4
    errs = []
    for N in (4, 8, 16):
        dt = 0.5 * (1.0 / N) / 2.0
        em = Eigenmode3DLF4(N, 2, dt, solver="explicit", output=False, hexahedral=True)
        em.elastic.u0.assign(em.elastic.u0.__class__(em.elastic.U).interpolate(em._u(0)))
        em.elastic.s0.assign(em.elastic.s0.__class__(em.elastic.S).interpolate(em._s(dt / 2.0)))
        u1, s1 = em.elastic.run(1.0)
        ue = em.elastic.u0.__class__(em.elastic.U).interpolate(em._u(1.0))
        se = em.elastic.s0.__class__(em.elastic.S).interpolate(em._s(1.0 + dt / 2.0))
        errs.append((np.abs(u1.dat.data_cells - ue.dat.data_cells).max(), np.abs(s1.dat.data_cells - se.dat.data_cells).max()))
    for k in (0, 1):
        assert math.log2(errs[0][k] / errs[1][k]) > 1.8 and math.log2(errs[1][k] / errs[2][k]) > 2.0, errs


@pytest.mark.parametrize("P,path", [(1, "generic"), (1, "lane"), (2, "generic"), (2, "lane"), (3, "generic"), (4, "generic"),
                                    (3, "hexm"), (4, "hexm")])
def test_sponge_source_and_material_on_hexahedra(gpu, monkeypatch, P, path):
    """The extras of the explosive-source set-up on hexahedral cells: DG4 sponge (elastic.py:207-208; 125 nodal values
    per cube), a nodal source table (:217-218) and per-cell lambda / mu, twelve steps against the oracle - on the
    table-driven generic kernels (what a DQ_1 / DQ_2 block this small runs by default), on the sum-factorised lane kernels
    and - degrees 3 and 4, their default at every size - on the plane-by-plane matrix kernels (kernels_hexm.hip)."""
    monkeypatch.setenv("SEIGEN_HIP_PATH", path)
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    n, L = (4, 3, 3), (2.0, 1.5, 1.2)
    h = [L[a] / n[a] for a in range(3)]
    nd = (P + 1) ** 3
    blk = HipBlock(3, P, n, h, [0.0] * 3, "quadrilateral")
    assert blk.nd == nd and blk.nfaces == 6 and blk.ncells == 36
    m = omesh.structured(3, n, L, quadrilateral=True)
    orc = OracleLF4(m, P)
    nc = m.ncells
    rng = np.random.default_rng(3)
    lam, mu = rng.uniform(0.4, 0.8, nc), rng.uniform(0.2, 0.4, nc)
    orc.dt, orc.l, orc.mu, orc.density = 0.02 * min(h) / P ** 2, lam, mu, 1.0
    X4 = m.node_coords(4)
    sigma = np.where(X4[..., 0] < 1.0, 30.0 * (1.0 - X4[..., 0]) * (1.0 + X4[..., 2]), 0.0)      # DG4 nodal values [nc, 125]
    orc.E.set_absorption(sigma, 4)
    nsteps = 12
    nodes = np.array([2 * nd + 1, 14 * nd + 3, 14 * nd + 4, 35 * nd + nd - 1])
    vals = rng.uniform(-1, 1, (nsteps, len(nodes), 3, 3))
    vals = 0.5 * (vals + np.swapaxes(vals, -1, -2))

    def source(k):
        S = np.zeros((nc * nd, 3, 3))
        S[nodes] = vals[k]
        return S.reshape(nc, nd, 3, 3)

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


@pytest.mark.parametrize("P", [1, 2])
@pytest.mark.parametrize("n", [(7, 5, 3), (70, 3, 2), (1, 1, 1), (64, 2, 1), (3, 9, 11)])
def test_hexahedral_lane_kernels_agree_with_the_generic_kernels(gpu, monkeypatch, P, n):
    """Ragged blocks (cell groups of 64 straddling rows and layers, a single cube, a row of exactly one group): the
    sum-factorised lane kernels against the table-driven generic kernels, with sponge, source, per-cell material and
    per-cell physical density, non-symmetric stress included (the lane kernels' symmetric-stress mode is left)."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    rng = np.random.default_rng(P * 100 + n[0])
    h = [0.7, 1.3, 0.9]
    for symmetric in (True, False):
        res = {}
        for path in ("generic", "lane"):
            monkeypatch.setenv("SEIGEN_HIP_PATH", path)
            blk = HipBlock(3, P, n, h, [0.0] * 3, "quadrilateral")
            nc, nd = blk.ncells, blk.nd
            if path == "generic":
                lam, mu, rho = rng.uniform(0.4, 0.8, nc), rng.uniform(0.2, 0.4, nc), rng.uniform(0.8, 1.6, nc)
                sigma = np.where(rng.uniform(size=(nc, 125)) > 0.7, 20.0, 0.0)
                nodes = np.unique(rng.integers(0, nc * nd, size=min(30, nc * nd)))
                vals = rng.uniform(-1, 1, (5, len(nodes), 3, 3))
                u0 = seeded(blk.field_shape(_lib.FIELD_U), 1)
                s0 = seeded(blk.field_shape(_lib.FIELD_S), 2)
                if symmetric:
                    vals = 0.5 * (vals + np.swapaxes(vals, -1, -2))
                    s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
            blk.set_params(1.0, 0.01 * min(h) / P ** 2, lam, mu)
            blk.set_density(rho, physical=True)
            blk.set_absorption(sigma, 4)
            blk.set_source(nodes, vals)
            blk.set_field(_lib.FIELD_U, u0)
            blk.set_field(_lib.FIELD_S, s0)
            blk.step(5)
            res[path] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S), blk.get_field(_lib.FIELD_UH),
                         blk.get_field(_lib.FIELD_SH))
            blk.close()
        for a, b in zip(res["lane"], res["generic"]):
            assert rel_err(a, b) < 1e-12


@pytest.mark.parametrize("P", [1, 2])
def test_hexahedral_lane_kernels_stage_parity_with_the_oracle(gpu, monkeypatch, P):
    """apply_F / apply_G (un-fused stage entry points) of the lane kernels against the oracle's assembled operators."""
    monkeypatch.setenv("SEIGEN_HIP_PATH", "lane")
    from oracle.forms import ElasticOperators
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    n, L = (3, 2, 4), (1.0, 1.5, 0.8)
    blk = HipBlock(3, P, n, [L[a] / n[a] for a in range(3)], [0.0] * 3, "quadrilateral")
    E = ElasticOperators(omesh.structured(3, n, L, quadrilateral=True), P)
    T = seeded(blk.field_shape(_lib.FIELD_S), 0)
    u = seeded(blk.field_shape(_lib.FIELD_U), 1)
    blk.set_params(1.0, 0.01, 0.7, 0.3)
    blk.set_field(_lib.FIELD_S, T)
    blk.set_field(_lib.FIELD_U, u)
    blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
    assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(T, u)) < 1e-12
    blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
    assert rel_err(blk.get_field(_lib.FIELD_SH), E.apply_G(u, 0.7, 0.3)) < 1e-12


@pytest.mark.parametrize("P", [3, 4])
@pytest.mark.parametrize("n", [(7, 5, 3), (35, 3, 2), (1, 1, 1), (16, 2, 1), (3, 4, 5)])
def test_hexahedral_matrix_kernels_agree_with_the_generic_kernels(gpu, monkeypatch, P, n):
    """DQ_3 / DQ_4 on the plane-by-plane matrix kernels (kernels_hexm.hip: 16 cubes per wave, x lines through
    v_mfma_f64_4x4x4, one component held, the rest streamed) against the thread-per-node generic kernels on ragged blocks -
    cell groups of 16 straddling rows and layers, a single cube, a row of exactly one group - with sponge, source,
    per-cell material and per-cell physical density; symmetric-stress storage and (a non-symmetric state) the full tensor."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    rng = np.random.default_rng(P * 100 + n[0])
    h = [0.7, 1.3, 0.9]
    for symmetric in (True, False):
        res = {}
        for path in ("generic", "hexm"):
            monkeypatch.setenv("SEIGEN_HIP_PATH", path)
            blk = HipBlock(3, P, n, h, [0.0] * 3, "quadrilateral")
            nc, nd = blk.ncells, blk.nd
            assert ("hexm_stage" in blk.stage_kernel_name(0)) == (path == "hexm")
            if path == "generic":
                lam, mu, rho = rng.uniform(0.4, 0.8, nc), rng.uniform(0.2, 0.4, nc), rng.uniform(0.8, 1.6, nc)
                sigma = np.where(rng.uniform(size=(nc, 125)) > 0.7, 20.0, 0.0)
                nodes = np.unique(rng.integers(0, nc * nd, size=min(30, nc * nd)))
                vals = rng.uniform(-1, 1, (5, len(nodes), 3, 3))
                u0 = seeded(blk.field_shape(_lib.FIELD_U), 1)
                s0 = seeded(blk.field_shape(_lib.FIELD_S), 2)
                if symmetric:
                    vals = 0.5 * (vals + np.swapaxes(vals, -1, -2))
                    s0 = 0.5 * (s0 + np.swapaxes(s0, -1, -2))
            blk.set_params(1.0, 0.01 * min(h) / P ** 2, lam, mu)
            blk.set_density(rho, physical=True)
            blk.set_absorption(sigma, 4)
            blk.set_source(nodes, vals)
            blk.set_field(_lib.FIELD_U, u0)
            blk.set_field(_lib.FIELD_S, s0)
            assert blk.is_sym() == (symmetric and path == "hexm")
            blk.step(5)
            res[path] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S), blk.get_field(_lib.FIELD_UH),
                         blk.get_field(_lib.FIELD_SH))
            blk.close()
        # two formulations of the same sums (the matrix kernels fold the own-trace half of the central flux into the line
        # operators, lift entries ~ (P + 1)^2; the generic kernel adds flux terms one by one) through five steps and the
        # derivative-like stage fields: round-off times the operators' amplification, measured 1e-11 .. 5e-11
        for a, b in zip(res["hexm"], res["generic"]):
            assert rel_err(a, b) < 2e-10


@pytest.mark.parametrize("P", [3, 4])
def test_hexahedral_matrix_kernels_stage_parity_with_the_oracle(gpu, P):
    """apply_F / apply_G (un-fused stage entry points) of the DQ_3 / DQ_4 matrix kernels against the oracle's assembled
    operators (seigen/elastic.py:204-219 by quadrature)."""
    from oracle.forms import ElasticOperators
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    n, L = (3, 2, 4), (1.0, 1.5, 0.8)
    blk = HipBlock(3, P, n, [L[a] / n[a] for a in range(3)], [0.0] * 3, "quadrilateral")
    assert "hexm_stage<%d, 0, 0, 1>" % P in blk.stage_kernel_name(0)
    E = ElasticOperators(omesh.structured(3, n, L, quadrilateral=True), P)
    T = seeded(blk.field_shape(_lib.FIELD_S), 0)
    T = 0.5 * (T + np.swapaxes(T, -1, -2))
    u = seeded(blk.field_shape(_lib.FIELD_U), 1)
    blk.set_params(1.0, 0.01, 0.7, 0.3)
    blk.set_field(_lib.FIELD_S, T)
    blk.set_field(_lib.FIELD_U, u)
    blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
    assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(T, u)) < 1e-12
    blk.apply_G(_lib.FIELD_U, _lib.FIELD_SH)
    assert rel_err(blk.get_field(_lib.FIELD_SH), E.apply_G(u, 0.7, 0.3)) < 1e-12


def test_hexahedral_blocks_equal_the_single_block_on_the_lane_kernels(gpu, monkeypatch):
    monkeypatch.setenv("SEIGEN_HIP_PATH", "lane")
    from tests.test_harness_gpu import _multiblock_case
    for pipelined in (True, False):
        _multiblock_case(3, 2, (4, 4, 4), (2, 2, 2), pipelined, extras=True, diagonal="quadrilateral")
        _multiblock_case(3, 1, (6, 3, 5), (3, 1, 2), pipelined, extras=True, diagonal="quadrilateral")
    _multiblock_case(3, 2, (130, 6, 2), (2, 3, 1), True, extras=True, separable=True, diagonal="quadrilateral")   # x sides, wide rows


def test_hexahedral_blocks_equal_the_single_block(gpu):
    """Blocks of hexahedral cells exchanging packed traces = the single block, bit for bit (the halo layer of
    elastic.py:404-436 is direction- and cell-type-agnostic): splits along every axis, sponge and source included."""
    from tests.test_harness_gpu import _multiblock_case
    for pipelined in (True, False):
        _multiblock_case(3, 2, (4, 4, 4), (2, 2, 2), pipelined, extras=True, diagonal="quadrilateral")
        _multiblock_case(3, 1, (6, 3, 5), (3, 1, 2), pipelined, extras=True, diagonal="quadrilateral")
    _multiblock_case(3, 2, (5, 6, 2), (1, 3, 1), True, diagonal="quadrilateral")
    _multiblock_case(3, 2, (6, 2, 3), (2, 1, 1), True, extras=True, separable=True, diagonal="quadrilateral")
    _multiblock_case(3, 3, (4, 2, 3), (2, 1, 3), True, extras=True, diagonal="quadrilateral")
    _multiblock_case(3, 4, (2, 4, 2), (1, 2, 2), False, extras=True, diagonal="quadrilateral")
    # DQ_3 / DQ_4 run the matrix kernels (kernels_hexm.hip): x sides with rows wider than a cell group, all three axes split
    _multiblock_case(3, 3, (36, 4, 2), (2, 2, 1), True, extras=True, diagonal="quadrilateral")
    _multiblock_case(3, 4, (6, 4, 4), (2, 2, 2), True, extras=True, separable=True, diagonal="quadrilateral")


def test_graph_replay_and_repeated_calls_on_hexahedra(gpu, monkeypatch):
    """hipGraph replay of whole steps (stages.cpp) on the hexahedral path equals the launch-by-launch run bit for bit,
    with a source, across several sg_step calls."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    res = {}
    for graph in (False, True):
        monkeypatch.setenv("SEIGEN_HIP_GRAPH", "1" if graph else "0")
        blk = HipBlock(3, 2, (3, 3, 2), [0.4, 0.3, 0.5], [0.0] * 3, "quadrilateral")
        blk.set_params(1.0, 0.002, 0.5, 0.25)
        blk.set_field(_lib.FIELD_U, seeded(blk.field_shape(_lib.FIELD_U), 1))
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 2)
        blk.set_field(_lib.FIELD_S, 0.5 * (s0 + np.swapaxes(s0, -1, -2)))
        r = np.random.default_rng(3)
        nodes = np.unique(r.integers(0, blk.ncells * blk.nd, size=20))
        sv = r.uniform(-1, 1, size=(25, len(nodes), 3, 3))
        blk.set_source(nodes, 0.5 * (sv + np.swapaxes(sv, -1, -2)))
        for c in (9, 1, 12):
            blk.step(c)
        res[graph] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    assert np.abs(res[True][0]).max() > 0


def test_unsupported_hexahedral_configurations_are_refused(gpu):
    from seigen_amd.backend import HipBlock
    with pytest.raises(Exception, match="degree"):
        HipBlock(3, 5, (2, 2, 2), [0.5] * 3, [0.0] * 3, "quadrilateral")
    with pytest.raises(Exception, match="f32"):
        HipBlock(3, 2, (2, 2, 2), [0.5] * 3, [0.0] * 3, "quadrilateral", dtype="f32")
