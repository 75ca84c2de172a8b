"""hipGraph replay of whole LF4 steps WITH a source (stages.cpp sg_step; kernels.hpp SrcStep).

The reference re-interpolates the source Expression before every step (seigen/elastic.py:285-288); on the device
that is a table of nodal values per step (or one slice and a weight per step).  A captured graph freezes kernel
arguments, so the replayed launches take the step index from a device word that a one-thread launch bumps at the end
of every step.  Replayed runs must equal the launch-by-launch runs (SEIGEN_HIP_GRAPH=0) bit for bit: per-step table,
separable source, static source, sources shorter than the run (the tail adds nothing), a run continued by a second
sg_step call, on the 2-D tile kernels (source fused into the G stages) and on the 3-D paths (source launches)."""
import numpy as np
import pytest

from tests.util import seeded

pytestmark = pytest.mark.gpu


def _run(monkeypatch, graph, dim, degree, n, kind, nsteps_src, calls):
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    monkeypatch.setenv("SEIGEN_HIP_GRAPH", "1" if graph else "0")
    h = [1.0 / n[a] for a in range(dim)]
    blk = HipBlock(dim, degree, n, h, [0.0] * dim)
    blk.set_params(1.0, 0.02 * min(h) / degree ** 2, 0.5, 0.25)
    blk.set_field(_lib.FIELD_U, seeded(blk.field_shape(_lib.FIELD_U), 1))
    s0 = seeded(blk.field_shape(_lib.FIELD_S), 2)
    blk.set_field(_lib.FIELD_S, 0.5 * (s0 + np.swapaxes(s0, -1, -2)))
    r = np.random.default_rng(3)
    nodes = np.unique(r.integers(0, blk.ncells * blk.nd, size=25))
    sv = r.uniform(-1, 1, size=(nsteps_src, len(nodes), dim, dim))
    sv = 0.5 * (sv + np.swapaxes(sv, -1, -2))
    if kind == "table":
        blk.set_source(nodes, sv)
    elif kind == "static":
        blk.set_source(nodes, sv[:1], static=True)
    else:
        blk.set_source_separable(nodes, sv[0], r.uniform(-2, 2, size=nsteps_src))
    for c in calls:
        blk.step(c)
    return blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S)


@pytest.mark.parametrize("dim,degree,n", [(2, 2, (12, 6)), (2, 3, (5, 4)), (3, 3, (3, 2, 2)), (3, 1, (3, 3, 2))])
@pytest.mark.parametrize("kind", ["table", "static", "separable"])
@pytest.mark.parametrize("nsteps_src,calls", [(30, (19,)), (7, (19,)), (30, (9, 1, 12)), (12, (10, 10))])
def test_graph_replay_with_source_is_bitwise_the_plain_run(gpu, monkeypatch, dim, degree, n, kind, nsteps_src, calls):
    ref = _run(monkeypatch, False, dim, degree, n, kind, nsteps_src, calls)
    got = _run(monkeypatch, True, dim, degree, n, kind, nsteps_src, calls)
    assert np.array_equal(got[0], ref[0])
    assert np.array_equal(got[1], ref[1])
    assert np.abs(ref[0]).max() > 0


def test_graph_replay_with_a_source_is_not_slower(gpu, monkeypatch):
    """With a source, 2000 steps of the reference's explosive-source mesh size (120 x 60 squares, P2) replayed from
    graphs must not take longer than launch by launch (measured: 23.7 against 27.5 us per step; the six dependent
    launches are bound by dispatch-to-dispatch latency on the device, so the gain is modest)."""
    import time
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    times = {}
    for graph in (False, True):
        monkeypatch.setenv("SEIGEN_HIP_GRAPH", "1" if graph else "0")
        blk = HipBlock(2, 2, (120, 60), [2.5, 2.5], [0.0, 0.0])
        blk.set_params(1.0, 1e-4, 3599.3664, 3600.0)
        nodes = np.arange(40, dtype=np.int64) + 6 * 2 * 50
        sv = np.zeros((2100, 40, 2, 2))
        sv[:, :, 0, 0] = sv[:, :, 1, 1] = np.sin(np.arange(2100))[:, None]
        blk.set_source(nodes, sv)
        blk.step(100)
        blk.sync()
        t0 = time.perf_counter()
        blk.step(2000)
        blk.sync()
        times[graph] = time.perf_counter() - t0
    assert times[True] < 1.05 * times[False], times


@pytest.mark.parametrize("dim,degree,n,diagonal", [(3, 3, (18, 3, 2), "left"), (3, 4, (17, 2, 2), "left"), (3, 3, (5, 3, 2), "quadrilateral"),
                                                   (3, 2, (6, 4, 3), "left")])
@pytest.mark.parametrize("calls", [(19,), (9, 1, 12), (1, 1, 8, 3)])
def test_graph_replay_with_a_sponge_pre_pass_is_bitwise_the_plain_run(gpu, monkeypatch, dim, degree, n, diagonal, calls):
    """Round 6: the sponge pre-pass of the 3-D families belongs to a STATE of the velocity field and is launched only when
    the buffer is stale (once per step, in stage UTEMP; stages.cpp fver / sponge_pre_ver).  That decision is made on the host
    while a graph is captured and replayed on the device: replayed runs (graphs of eight steps and of one, runs continued by
    further sg_step calls, an upload of the velocity in between) must equal the launch-by-launch runs bit for bit - with cells
    of every kind: no sponge, constant, affine, general."""
    from seigen_amd import _lib
    from seigen_amd.backend import HipBlock
    from tests.util import oracle_mesh
    L = tuple(0.4 * k for k in n)
    h = [L[a] / n[a] for a in range(dim)]
    Xq = oracle_mesh(dim, n, L, diagonal).node_coords(4)
    r = np.random.default_rng(5)
    kind = r.integers(0, 4, size=Xq.shape[0])
    sigma = np.zeros(Xq.shape[:2])
    sigma[kind == 1] = 7.0
    sigma[kind == 2] = (3.0 + 11.0 * Xq[..., 0] + 7.0 * Xq[..., 1] + 5.0 * Xq[..., 2])[kind == 2]
    sigma[kind == 3] = r.uniform(0.0, 20.0, size=(int((kind == 3).sum()), Xq.shape[1]))
    res = {}
    for graph in (False, True):
        monkeypatch.setenv("SEIGEN_HIP_GRAPH", "1" if graph else "0")
        blk = HipBlock(dim, degree, n, h, [0.0] * dim, diagonal)
        blk.set_params(1.0, 0.02 * min(h) / degree ** 2, 0.5, 0.25)
        u0 = seeded(blk.field_shape(_lib.FIELD_U), 1)
        blk.set_field(_lib.FIELD_U, u0)
        s0 = seeded(blk.field_shape(_lib.FIELD_S), 2)
        blk.set_field(_lib.FIELD_S, 0.5 * (s0 + np.swapaxes(s0, -1, -2)))
        blk.set_absorption(sigma, 4)
        for i, c in enumerate(calls):
            blk.step(c)
            if i == 0 and len(calls) > 2:      # the velocity is replaced between two calls: the buffer's state is stale
                blk.set_field(_lib.FIELD_U, 0.5 * blk.get_field(_lib.FIELD_U) + 0.1 * u0)
        res[graph] = (blk.get_field(_lib.FIELD_U), blk.get_field(_lib.FIELD_S))
        blk.close()
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    assert np.isfinite(res[True][0]).all() and np.abs(res[True][0]).max() > 0
