"""BASELINE configs 2, 3 and 5 at FULL size, production kernels against the ORACLE: the committed goldens of
tests/golden/make_golden_fullsize.py (the oracle's plain-C restatement, oracle/c/seigen_oracle.c, run on the same
inputs: tests/fullsize_cases.py).  Compared: ~400 cells sampled over the whole mesh (corners, source cells,
sponge cells, random interior) in every field the product keeps - state u, s and the stage fields utemp, sh1 of
the last step - plus the sums of every field over each slab of the mesh (covers every cell).  State to 1e-10 of
its scale; the stage fields are derivatives of the state and carry the operators' round-off amplification
(tests/test_fullsize_gpu.py): 1e-9 / 1e-7 in 3-D at P4.
Follows seigen/elastic.py:204-219 (forms), :340-352 (combines), :285-288 (source), :207-208 (sponge)."""
import os

import numpy as np
import pytest

from tests import fullsize_cases as fc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _quiet():
    import seigen_amd
    import seigen_amd.helpers as helpers
    helpers.log = lambda s: None
    seigen_amd.elastic.log = lambda s: None


def _compare(blk, gold, nlayers, tol, dt):
    """State and work fields after the run against the oracle's golden.  The UH buffer holds, after a step, not the
    reference's `utemp` but w = dt u1 + dt^3/24 utemp - the one velocity the stress update needs (g is linear:
    dt sh1 + dt^3/24 sh2 = G(w), csrc/stages.cpp) - so it is compared with that combination of the golden's u and utemp,
    its error measured in units of what the two tolerances allow: dt tol_u |u| + dt^3/24 tol_uh |utemp|."""
    from seigen_amd import _lib
    fields = (("u", _lib.FIELD_U), ("s", _lib.FIELD_S), ("uh", _lib.FIELD_UH), ("sh", _lib.FIELD_SH))
    cells = gold["cells"]
    c3 = dt ** 3 / 24.0
    worst = {}
    for name, f in fields:
        full = blk.get_field(f)
        want, want_layers = gold[name], gold[name + "_layers"]
        scale = np.abs(want).max()
        lscale = np.abs(want_layers).max()
        unit = 1.0
        if name == "uh":
            su, suh = np.abs(gold["u"]).max(), np.abs(gold["uh"]).max()
            unit = (dt * tol["u"] * su + c3 * tol["uh"] * suh) / tol["uh"]      # err / unit < tol["uh"] <=> within both allowances
            want, want_layers = dt * gold["u"] + c3 * gold["uh"], dt * gold["u_layers"] + c3 * gold["uh_layers"]
            scale = lscale = unit
        assert scale > 0 and np.isfinite(full).all()
        err = np.abs(full[cells] - want).max() / scale
        lay = fc.layer_sums(full, nlayers)
        lerr = np.abs(lay - want_layers).max() / max(lscale, scale)
        worst[name] = (err, lerr)
        # a slab sum adds up to (cells per slab) values: allow sqrt-like growth of round-off
        assert err < tol[name] and lerr < 30 * tol[name], (name, err, lerr)
        del full
    return worst


def operator_amplification(dim, degree, h, diagonal=0):
    """A = max over classes of  sum_r |Jinv_r.|_1 |||D_r|||_inf + sum_f |(c n)_f|_1 |||L_f|||_inf  - the infinity norm
    of the element operator behind `f` and `g` (seigen/elastic.py:204-219) with absolute values taken entry by entry,
    from the library's device-free tables.  One application computed in floating point differs from the exact one
    by at most n eps A max|input| (n = number of terms of a row), and an input that is off by delta is answered with
    at most A delta: the round-off amplification of a stage field, computed instead of narrated."""
    import ctypes as C
    from seigen_amd import _lib
    lib = _lib.load()
    nd = int(round(lib.sg_reference_operator(dim, degree, 2, 0, None, 0) ** 0.5))

    def op(which):
        n = lib.sg_reference_operator(dim, degree, which, 0, None, 0)
        out = np.empty(n)
        assert lib.sg_reference_operator(dim, degree, which, 0, out.ctypes.data, out.nbytes) == n
        return out
    D = op(0).reshape(dim, nd, nd)
    L = op(1).reshape(dim + 1, nd, -1)
    nf = L.shape[2]
    hh = (C.c_double * 3)(*[h[a] if a < dim else 1.0 for a in range(3)])
    nb, nbn = np.zeros((6, 4, 5), dtype=np.int32), np.zeros((6, 4, nf), dtype=np.int32)
    cn, jinv = np.zeros((6, 4, 3)), np.zeros((6, 3, 3))
    assert lib.sg_mesh_tables(dim, degree, diagonal, hh, nb.ctypes.data, nbn.ctypes.data, cn.ctypes.data, jinv.ctypes.data) == 0
    nD = [np.abs(D[r]).sum(axis=1).max() for r in range(dim)]
    nL = [np.abs(L[f]).sum(axis=1).max() for f in range(dim + 1)]
    ncls = {1: 1, 2: 2, 3: 6}[dim]
    return max(sum(np.abs(jinv[k][r]).sum() * nD[r] for r in range(dim)) +
               sum(np.abs(cn[k][f]).sum() * nL[f] for f in range(dim + 1)) for k in range(ncls))


def test_config3_full_size_vs_oracle(gpu):
    """3-D eigenmode, 64^3 cubes x 6 tets, P4 (BASELINE config 3, the bench workload): three LF4 steps on the MFMA
    kernels from the product's own nodal interpolation of the eigenmode (bench.fill_initial_condition)."""
    _quiet()
    import bench
    from seigen_amd import BoxMesh, ElasticLF4
    c = fc.C3
    gold = np.load(os.path.join(GOLD, "fullsize_c3.npz"))
    mesh = BoxMesh(c["n"], c["n"], c["n"], 1.0, 1.0, 1.0)
    el = ElasticLF4.create(mesh, "DG", c["P"], dimension=3, solver="explicit", output=False)
    el.density, el.mu, el.l, el.dt = c["rho"], c["mu"], c["lam"], c["dt"]
    bench.fill_initial_condition(el, el.dt)
    el.setup()
    blk = el.block
    blk.set_source([], None)
    blk.step(c["steps"])
    worst = _compare(blk, gold, c["n"], dict(u=1e-10, s=1e-10, sh=1e-9, uh=1e-7), c["dt"])
    # The literals above are what is observed (a regression guard).  What MUST hold is computed: the stage fields are
    # operator applications of the state - sh1 = g(u1), utemp = f(sh1) (elastic.py:298-303) - so their errors are at
    # most the operator's amplification A (about 1.2e5 here: |||D_r|||_inf = 232 at P4, 1/h = 64) times the error of
    # their input, plus the round-off of one application; relative to the fields' own scales:
    A = operator_amplification(3, c["P"], [1.0 / c["n"]] * 3)
    eps, nterms = np.finfo(np.float64).eps, 3 * 3 * 35 + 4 * 3 * 15
    scale = {k: float(np.abs(gold[k]).max()) for k in ("u", "s", "uh", "sh")}
    lame = c["lam"] + 2 * c["mu"]                                  # g scales its operator by lambda, mu
    bound_sh = lame * A * scale["u"] * (worst["u"][0] + nterms * eps) / scale["sh"]
    bound_uh = A * scale["sh"] * (worst["sh"][0] + nterms * eps) / scale["uh"]
    assert worst["sh"][0] <= bound_sh, (worst, bound_sh)
    # (UH holds w = dt u1 + dt^3/24 utemp: its error is at most dt x u's + dt^3/24 x utemp's, in _compare's unit for "uh")
    dt, c3 = c["dt"], c["dt"] ** 3 / 24.0
    unit = (dt * 1e-10 * scale["u"] + c3 * 1e-7 * scale["uh"]) / 1e-7
    bound_w = (dt * (worst["u"][0] + eps) * scale["u"] + c3 * bound_uh * scale["uh"]) / unit
    assert worst["uh"][0] <= bound_w, (worst, bound_w)
    assert 1e5 < A < 2e5
    blk.close()


def _box_source_expression(lo, hi, t0):
    from seigen_amd import Expression
    box = "x[0] >= %r && x[0] <= %r && x[1] >= %r && x[1] <= %r" % (lo[0], hi[0], lo[1], hi[1])
    code = "%s ? (-1.0 + 2*a*pow(t - t0, 2))*exp(-a*pow(t - t0, 2)) : 0.0" % box
    return Expression(((code, "0.0"), ("0.0", code)), a=fc.A_RICKER, t0=t0, t=0)


@pytest.mark.parametrize("quadrilateral", [False, True])
def test_config2_full_size_vs_oracle(gpu, quadrilateral):
    """2-D explosive source, 512 x 512 squares, P2, DG4 sponge + box-Ricker source (BASELINE config 2) through the
    solver class on the 2-D MFMA tile kernels, 20 steps from a smooth state - on triangles (the reference's mesh) and
    on quadrilateral cells (DQ_2; golden fullsize_c2q.npz from the same C port of the oracle)."""
    _quiet()
    from seigen_amd import ElasticLF4, Expression, Function, FunctionSpace, RectangleMesh
    c = fc.C2
    gold = np.load(os.path.join(GOLD, "fullsize_c2q.npz" if quadrilateral else "fullsize_c2.npz"))
    n, h = c["n"], c["h"]
    L = n * h
    mesh = RectangleMesh(n, n, L, L, quadrilateral=quadrilateral)
    el = ElasticLF4.create(mesh, "DG", c["P"], dimension=2, solver="explicit", output=False)
    el.density, el.mu, el.l, el.dt = c["rho"], c["mu"], c["lam"], c["dt"]
    sx, sy, hw = c["src"][0], c["src"][1], c["src_half"]
    el.source_expression = _box_source_expression((sx - hw, sy - hw), (sx + hw, sy + hw), 10 * c["dt"])
    el.source_function = Function(el.S)
    el.source = el.source_expression
    el.absorption_function = Function(FunctionSpace(mesh, "DG", c["sigma_degree"]))
    el.absorption = Expression("x[0] <= %r || x[0] >= %r || x[1] <= %r ? %r : 0" % (c["sponge"], L - c["sponge"], c["sponge"], c["sigma"]))
    u0, s0 = fc.smooth_state(el.U.node_coords(), c["k"], c["s_scale"])
    el.u0.assign(Function(el.U).assign(u0))
    el.s0.assign(Function(el.S).assign(s0))
    el.run(c["steps"] * c["dt"] * (1 + 1e-9))
    assert el.block.counters()["steps"] == c["steps"]
    _compare(el.block, gold, n, dict(u=1e-10, s=1e-10, sh=1e-9, uh=1e-8), c["dt"])


def test_config5_full_size_vs_oracle(gpu):
    """Marmousi 383 x 121 squares, P3, per-cell lambda / mu and Gardner density (physical update), box-Ricker
    source (BASELINE config 5) on the 2-D MFMA tile kernels, 20 steps from a smooth state."""
    _quiet()
    from seigen_amd import ElasticLF4, Function, RectangleMesh
    from seigen_amd.marmousi import NX, NY, H, cell_material, gardner_density
    c = fc.C5
    gold = np.load(os.path.join(GOLD, "fullsize_c5.npz"))
    nx, ny = NX - 1, NY - 1
    mesh = RectangleMesh(nx, ny, nx * H, ny * H)
    el = ElasticLF4.create(mesh, "DG", c["P"], dimension=2, solver="explicit", output=False)
    lam, mu, vp = cell_material(el.U, density=gardner_density)
    rho = gardner_density(vp)
    assert fc.digest(lam, mu, rho) == str(gold["material_digest"]), "the material arrays are not the golden's"
    dt = c["courant"] * H / float(vp.max())
    assert dt == float(gold["dt"])
    el.density, el.density_physical, el.l, el.mu, el.dt = rho, True, lam, mu, dt
    sx, sy, hw = 0.5 * nx * H, ny * H - 24.0, c["src_half"]
    el.source_expression = _box_source_expression((sx - hw, sy - hw), (sx + hw, sy + hw), 10 * dt)
    el.source_function = Function(el.S)
    el.source = el.source_expression
    u0, s0 = fc.smooth_state(el.U.node_coords(), c["k"], c["s_scale"])
    el.u0.assign(Function(el.U).assign(u0))
    el.s0.assign(Function(el.S).assign(s0))
    el.run(c["steps"] * dt * (1 + 1e-9))
    assert el.block.counters()["steps"] == c["steps"]
    _compare(el.block, gold, ny, dict(u=1e-10, s=1e-10, sh=1e-9, uh=1e-8), dt)
