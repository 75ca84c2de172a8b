import sys, os, time
sys.path.insert(0, os.getcwd())
import seigen_amd, seigen_amd.helpers as helpers
import seigen_amd.harness.explosive_source as hx
import seigen_amd.harness.eigenmode as he
helpers.log = seigen_amd.elastic.log = hx.log = he.log = lambda s: None
from seigen_amd.harness.explosive_source import ExplosiveSourceLF4
for quad in (False, True, False, True):      # the second pair runs warm (code objects loaded, allocator primed)
    es = ExplosiveSourceLF4()
    t0 = time.perf_counter()
    el = es.setup(300.0, 150.0, 2.5, degree=2, dt=0.001, quadrilateral=quad)
    t1 = time.perf_counter()
    u1, s1 = el.run(2.5)
    el.block.sync()
    t2 = time.perf_counter()
    c = el.block.counters()
    print("explosive source 120x60 P2 quad=%s: setup %.3f s, run(T=2.5) %.3f s for %d steps = %.1f us/step wall" % (quad, t1 - t0, t2 - t1, c["steps"], (t2 - t1) / c["steps"] * 1e6), flush=True)
for (N, P) in ((32, 4), (40, 1)):
    dt = 0.5 * (1.0 / N) / 2.0 ** (P - 1)
    t0 = time.perf_counter()
    em = he.Eigenmode2DLF4(N, P, dt, output=False)
    u1, s1 = em.eigenmode2d(T=5.0)
    e = em.eigenmode_error(u1, s1)
    t1 = time.perf_counter()
    print("eigenmode 2-D N=%d P=%d: %.3f s end to end (errors %.3e %.3e), %d steps" % (N, P, t1 - t0, e[0], e[1], em.elastic.block.counters()["steps"]), flush=True)
t0 = time.perf_counter()
em = he.Eigenmode3DLF4(8, 3, 0.5 / 8 / 4, output=False)
u1, s1 = em.eigenmode3d(T=5.0)
e = em.eigenmode_error(u1, s1)
print("eigenmode 3-D N=8 P=3: %.3f s end to end, %d steps" % (time.perf_counter() - t0, em.elastic.block.counters()["steps"]), flush=True)
