"""The oracle's plain-C restatement (oracle/c/seigen_oracle.c, the CPU baseline of bench.py)
against the numpy oracle."""
import numpy as np
import pytest

from oracle import mesh as omesh
from oracle.cport import CPort
from oracle.lf4 import OracleLF4


@pytest.mark.parametrize("dim,n,P", [(1, (6,), 2), (2, (3, 4), 3), (3, (2, 2, 2), 2), (3, (2, 1, 2), 4)])
def test_cport_matches_numpy_oracle(dim, n, P):
    m = omesh.structured(dim, n, tuple(1.0 + 0.25 * a for a in range(dim)))
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
