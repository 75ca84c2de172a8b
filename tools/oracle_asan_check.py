#!/usr/bin/env python
"""The oracle's C port (oracle/c/seigen_oracle.c) under AddressSanitizer + UBSan (SURVEY 5 "ASAN on CPU restatement").

Run by tests/test_host_asan.py as
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
    SEIGEN_ORACLE_LIB=oracle/c/libseigen_oracle_asan.so python tools/oracle_asan_check.py
(the interpreter itself is not instrumented, hence no leak check): LF4 steps of a 3^3-cube P4 mesh and of 2-D meshes
with sponge, source, per-cell material and density - every code path of so_step / so_step_ex / so_apply_* - compared
with the numpy oracle so that the run is also a correctness check of the instrumented build."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mesh as omesh  # noqa: E402
from oracle.cport import CPort, sponge_blocks  # noqa: E402
from oracle.forms import ElasticOperators  # noqa: E402
from oracle.lf4 import OracleLF4  # noqa: E402


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def main():
    assert os.environ.get("SEIGEN_ORACLE_LIB", "").endswith("_asan.so"), "run with SEIGEN_ORACLE_LIB=<the asan build>"
    rng = np.random.default_rng(5)
    # 3-D, 3^3 cubes x 6 tets, P4: so_step against the numpy oracle
    m = omesh.UnitCubeMesh(3, 3, 3)
    cp = CPort(m, 4)
    orc = OracleLF4(m, 4)
    orc.dt, orc.l, orc.mu, orc.density = 1e-3, 0.5, 0.25, 1.0
    orc.u0 = rng.uniform(-1, 1, orc.u0.shape)
    orc.s0 = rng.uniform(-1, 1, orc.s0.shape)
    orc.s0 = 0.5 * (orc.s0 + orc.s0.swapaxes(-1, -2))
    u, s = cp.step(orc.u0, orc.s0, 1.0, orc.dt, orc.l, orc.mu, 2)
    for k in range(2):
        orc.step((k + 1) * orc.dt)
    assert rel(u, orc.u1) < 1e-11 and rel(s, orc.s1) < 1e-11, (rel(u, orc.u1), rel(s, orc.s1))
    E = ElasticOperators(m, 4)
    assert rel(cp.apply_F(orc.s0), E.apply_F(orc.s0, None)) < 1e-11
    assert rel(cp.apply_G(orc.u0, 0.7, 0.3), E.apply_G(orc.u0, 0.7, 0.3)) < 1e-11
    for threads in (1, 3):
        cp.set_threads(threads)
        u2, s2 = cp.step(orc.u0, orc.s0, 1.0, orc.dt, orc.l, orc.mu, 1)
        assert np.isfinite(u2).all() and np.isfinite(s2).all()
    # 2-D, triangles and quadrilaterals, P2: sponge + source + per-cell material + physical density (so_step_ex)
    for quad in (False, True):
        m2 = omesh.RectangleMesh(5, 4, 5.0, 4.0, quadrilateral=quad)
        c2 = CPort(m2, 2)
        nc, nd = m2.ncells, c2.nd
        Xs = m2.node_coords(4)
        sig = np.where(Xs[..., 0] <= 1.0, 10.0, 0.0)
        nodes = np.array([3, nd + 1, 2 * nd + 2], dtype=np.int64)
        vals = rng.uniform(-1, 1, (3, len(nodes), 2, 2))
        c2.set_extra(lam=rng.uniform(0.4, 0.6, nc), mu=rng.uniform(0.2, 0.3, nc), rho=rng.uniform(0.9, 1.1, nc),
                     rho_physical=True, sponge=sponge_blocks(m2, 2, sig, 4), src_nodes=nodes, src_values=vals)
        u0 = rng.uniform(-1, 1, (nc, nd, 2))
        s0 = rng.uniform(-1, 1, (nc, nd, 2, 2))
        u3, s3 = c2.step_ex(u0, s0, 1.0, 1e-3, 0.0, 0.0, 5)       # two steps beyond the source table's three
        assert np.isfinite(u3).all() and np.isfinite(s3).all()
    print("oracle_asan_check: clean")


if __name__ == "__main__":
    main()
