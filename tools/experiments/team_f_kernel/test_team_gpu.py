"""The trace-sharing F kernels (kernels_mfma.hip mfma_stage_FT, SEIGEN_HIP_TEAM = 4 / 8; double, 3-D, degrees 3
and 4): teams of waves exchange the intra-cube facet traces of `f` (seigen/elastic.py:204-209, the `dS` term :206)
through LDS instead of re-reading the neighbours' tensors.  Off by default (profiles/r04/team_f_kernel.txt: fewer
bytes, more time); kept as a measured alternative and tested like the production kernels:
  * against the CPU oracle per operator application and over whole steps,
  * against the plain kernels on ragged blocks (same sums, another facet order: round-off only),
  * multi-block = single-block BITWISE - a facet's trace comes from a team mate's stash or from memory depending on
    how the items of a launch fall on the teams, and the two must give the same bits,
  * with a sponge, a source, per-cell material, and non-symmetric stress (the full-tensor instantiation)."""
import numpy as np
import pytest

from oracle.forms import ElasticOperators
from oracle.lf4 import OracleLF4
from tests.util import oracle_mesh, rel_err, seeded

pytestmark = pytest.mark.gpu

CASES = [
    (3, (2, 2, 2), (1.0, 1.0, 1.0)),
    (4, (3, 1, 2), (1.0, 1.0, 1.0)),
    (3, (5, 3, 17), (1.0, 0.6, 3.4)),     # chunked item order, groups straddling rows and layers
    (4, (3, 2, 19), (0.6, 0.4, 3.8)),
    (4, (17, 3, 2), (1.7, 0.3, 0.2)),     # a second cell group in every row
]


def make_block(degree, n, L, **kw):
    from seigen_amd.backend import HipBlock
    return HipBlock(3, degree, n, [L[a] / n[a] for a in range(3)], [0.0] * 3, "left", **kw)


@pytest.mark.parametrize("team", [4, 8])
@pytest.mark.parametrize("degree,n,L", CASES)
def test_team_apply_F_and_steps_vs_oracle(gpu, monkeypatch, team, degree, n, L):
    from seigen_amd import _lib
    monkeypatch.setenv("SEIGEN_HIP_TEAM", str(team))
    blk = make_block(degree, n, L)
    m = oracle_mesh(3, n, L)
    E = ElasticOperators(m, degree)
    for sym in (True, False):      # symmetric storage, then the full-tensor kernels (a non-symmetric upload leaves sym mode)
        T = seeded(blk.field_shape(_lib.FIELD_S), 10 + team)
        if sym:
            T = 0.5 * (T + np.swapaxes(T, -1, -2))
        u = seeded(blk.field_shape(_lib.FIELD_U), 11)
        blk.set_params(1.0, 0.01, 0.7, 0.3)
        blk.set_field(_lib.FIELD_S, T)
        blk.set_field(_lib.FIELD_U, u)
        blk.apply_F(_lib.FIELD_S, _lib.FIELD_U, _lib.FIELD_UH)
        assert rel_err(blk.get_field(_lib.FIELD_UH), E.apply_F(T, u)) < 1e-11
    orc = OracleLF4(m, degree)
    hmin = min(L[a] / n[a] for a in range(3))
    orc.dt, orc.l, orc.mu, orc.density = 0.05 * hmin / degree ** 2, 0.5, 0.25, 1.0
    orc.u0 = seeded(blk.field_shape(_lib.FIELD_U), 2)
    orc.s0 = seeded(blk.field_shape(_lib.FIELD_S), 3)
    blk.set_params(orc.density, orc.dt, orc.l, orc.mu)
    blk.set_field(_lib.FIELD_U, orc.u0)
    blk.set_field(_lib.FIELD_S, orc.s0)
    blk.step(3)
    for k in range(3):
        orc.step((k + 1) * orc.dt)
    assert rel_err(blk.get_field(_lib.FIELD_U), orc.u1) < 1e-10
    assert rel_err(blk.get_field(_lib.FIELD_S), orc.s1) < 1e-10


@pytest.mark.parametrize("team", [4, 8])
def test_team_kernels_against_plain_kernels(gpu, monkeypatch, team):
    """Same block, sponge + source + per-cell material, plain and team F kernels: equal to round-off (the team kernels
    sum the facets in the order 0, 3, 1, 2)."""
    from seigen_amd import _lib
    degree, n, L = 4, (20, 5, 6), (2.0, 0.5, 0.6)
    out = {}
    for t in (0, team):
        monkeypatch.setenv("SEIGEN_HIP_TEAM", str(t))
        blk = make_block(degree, n, L)
        r = np.random.default_rng(5)
        lam = r.uniform(0.4, 0.8, blk.ncells)
        mu = r.uniform(0.2, 0.4, blk.ncells)
        blk.set_params(1.0, 1e-3, lam, mu)
        blk.set_field(_lib.FIELD_U, seeded(blk.field_shape(_lib.FIELD_U), 6))
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 7)
        blk.set_field(_lib.FIELD_S, 0.5 * (s0 + np.swapaxes(s0, -1, -2)))
        blk.set_absorption(np.where(r.uniform(size=(blk.ncells, 35)) > 0.7, 2.0, 0.0), 4)
        nodes = np.unique(r.integers(0, blk.ncells * blk.nd, size=30))
        sv = r.uniform(-1, 1, size=(4, len(nodes), 3, 3))
        blk.set_source(nodes, 0.5 * (sv + np.swapaxes(sv, -1, -2)))
        blk.step(4)
        out[t] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
    assert rel_err(out[team][0], out[0][0]) < 1e-12
    assert rel_err(out[team][1], out[0][1]) < 1e-12


@pytest.mark.parametrize("team", [4, 8])
@pytest.mark.parametrize("degree,n,grid", [
    (4, (4, 2, 4), (2, 1, 2)),
    (3, (9, 9, 9), (3, 3, 3)),        # centre block: seven-box regions
    (4, (120, 2, 2), (3, 1, 1)),      # x sides: group-thick shells, item lists
    (3, (40, 3, 4), (2, 1, 2)),
])
def test_team_multiblock_bitwise(gpu, monkeypatch, team, degree, n, grid):
    from tests.test_harness_gpu import _multiblock_case
    monkeypatch.setenv("SEIGEN_HIP_TEAM", str(team))
    for pipelined in (True, False):
        _multiblock_case(3, degree, n, grid, pipelined, extras=True)
