"""The oracle's plain-C restatement (oracle/c/seigen_oracle.c, the CPU baseline of bench.py)
against the numpy oracle."""
import numpy as np
import pytest

from oracle import mesh as omesh
from oracle.cport import CPort
from oracle.lf4 import OracleLF4


@pytest.mark.parametrize("dim,n,P,quad", [(1, (6,), 2, False), (2, (3, 4), 3, False), (3, (2, 2, 2), 2, False),
                                          (3, (2, 1, 2), 4, False), (2, (4, 3), 2, True), (2, (3, 3), 4, True),
                                          (3, (3, 2, 2), 2, True), (3, (2, 2, 1), 3, True), (3, (1, 2, 2), 4, True)])
def test_cport_matches_numpy_oracle(dim, n, P, quad):
    m = omesh.structured(dim, n, tuple(1.0 + 0.25 * a for a in range(dim)), quadrilateral=quad)
    orc = OracleLF4(m, P)
    cp = CPort(m, P)
    rng = np.random.default_rng(0)
    u = rng.uniform(-1, 1, (m.ncells, orc.E.nd, dim))
    T = rng.uniform(-1, 1, (m.ncells, orc.E.nd, dim, dim))
    assert np.abs(cp.apply_F(T) - orc.E.apply_F(T)).max() < 1e-11 * np.abs(orc.E.apply_F(T)).max()
    G = orc.E.apply_G(u, 0.7, 0.3)
    assert np.abs(cp.apply_G(u, 0.7, 0.3) - G).max() < 1e-11 * np.abs(G).max()
    orc.u0, orc.s0 = u.copy(), T.copy()
    orc.dt, orc.l, orc.mu, orc.density = 1e-3, 0.5, 0.25, 1.0
    for k in range(3):
        orc.step((k + 1) * orc.dt)
    cu, cs = cp.step(u, T, 1.0, orc.dt, 0.5, 0.25, 3)
    assert np.abs(cu - orc.u1).max() < 1e-10 * np.abs(orc.u1).max()
    assert np.abs(cs - orc.s1).max() < 1e-10 * np.abs(orc.s1).max()
    assert cp.threads() >= 1


def test_cport_extras_match_numpy_oracle():
    """so_step_ex: sponge (DG4 sigma), time-dependent stress source, per-cell lambda / mu / density in both
    update rules - the ingredients of BASELINE configs 2, 4 and 5 - against the numpy oracle, 4 steps."""
    for dim, n, P, physical, quad in ((2, (6, 5), 2, False, False), (2, (5, 4), 3, True, False), (3, (3, 2, 2), 4, False, False),
                                      (2, (6, 5), 2, True, True), (3, (3, 2, 2), 2, False, True), (3, (2, 1, 2), 3, True, True)):
        m = omesh.structured(dim, n, tuple(2.0 + 0.5 * a for a in range(dim)), quadrilateral=quad)
        orc = OracleLF4(m, P)
        cp = CPort(m, P)
        rng = np.random.default_rng(5)
        nd = orc.E.nd
        u = rng.uniform(-1, 1, (m.ncells, nd, dim))
        T = rng.uniform(-1, 1, (m.ncells, nd, dim, dim))
        T = T + np.swapaxes(T, 2, 3)
        lam = rng.uniform(0.4, 0.9, m.ncells)
        mu = rng.uniform(0.2, 0.5, m.ncells)
        rho = rng.uniform(0.8, 1.5, m.ncells)
        Xs = m.node_coords(4)
        sig = np.where(Xs[..., 0] < 0.7, 40.0 * (1.0 + Xs[..., 0]), 0.0)          # sigma varies inside the sponge cells
        orc.E.set_absorption(sig, 4)
        src_nodes = np.sort(rng.choice(m.ncells * nd, 7, replace=False))
        nsteps = 4
        vals = rng.uniform(-1, 1, (nsteps, 7, dim, dim))
        vals = vals + np.swapaxes(vals, 2, 3)

        def source(t, dt=1e-3):
            k = int(round(t / dt)) - 1
            S = np.zeros((m.ncells * nd, dim, dim))
            S[src_nodes] = vals[k]
            return S.reshape(m.ncells, nd, dim, dim)

        orc.source = source
        orc.u0, orc.s0 = u.copy(), T.copy()
        orc.dt, orc.l, orc.mu, orc.density, orc.density_physical = 1e-3, lam, mu, rho, physical
        for k in range(nsteps):
            orc.step((k + 1) * orc.dt)
        cp.set_extra(lam=lam, mu=mu, rho=rho, rho_physical=physical, absorb=orc.E.absorb, src_nodes=src_nodes, src_values=vals)
        cu, cs = cp.step_ex(u, T, 1.0, orc.dt, 0.0, 0.0, nsteps)
        assert np.abs(cu - orc.u1).max() < 1e-11 * np.abs(orc.u1).max(), (dim, P)
        assert np.abs(cs - orc.s1).max() < 1e-11 * np.abs(orc.s1).max(), (dim, P)
        # the sponge blocks built for the sigma-carrying cells only (full-size goldens) = the numpy oracle's
        from oracle.cport import sponge_blocks
        cp.set_extra(lam=lam, mu=mu, rho=rho, rho_physical=physical, sponge=sponge_blocks(m, P, sig, 4), src_nodes=src_nodes,
                     src_values=vals)
        cu3, cs3 = cp.step_ex(u, T, 1.0, orc.dt, 0.0, 0.0, nsteps)
        assert np.abs(cu3 - cu).max() < 1e-13 * np.abs(cu).max() and np.abs(cs3 - cs).max() < 1e-13 * np.abs(cs).max()
        # split in two calls: the source index continues
        cu2, cs2 = cp.step_ex(u, T, 1.0, orc.dt, 0.0, 0.0, 2)
        cu2, cs2 = cp.step_ex(cu2, cs2, 1.0, orc.dt, 0.0, 0.0, 2, step0=2)
        assert np.array_equal(cu2, cu3) and np.array_equal(cs2, cs3)
