"""1-D Gaussian pulse with sponge ends, as ``tests/pulse/pulse_1d_lf4.py`` of the reference."""
from seigen_amd import ElasticLF4, Expression, Function, FunctionSpace, IntervalMesh, Vp, Vs, log, timed_region


def pulse_1d_lf4(T=2.0, Lx=4.0, h=1e-2, dt=0.0025, degree=1, solver="explicit", output=False):
    with timed_region('mesh generation'):
        mesh = IntervalMesh(int(Lx/h), Lx)
    elastic = ElasticLF4.create(mesh, "DG", degree, dimension=1, solver=solver, output=output)

    # Constants (pulse_1d_lf4.py:13-17)
    elastic.density = 1.0
    elastic.dt = dt
    elastic.mu = 0.25
    elastic.l = 0.5

    log("P-wave velocity: %f" % Vp(elastic.mu, elastic.l, elastic.density))
    log("S-wave velocity: %f" % Vs(elastic.mu, elastic.density))

    # sponge at both ends (pulse_1d_lf4.py:22-24)
    F = FunctionSpace(elastic.mesh, "DG", 1)
    elastic.absorption_function = Function(F)
    elastic.absorption = Expression("x[0] >= %r || x[0] <= 0.5 ? 100.0 : 0" % (Lx - 0.5))

    # Initial conditions (pulse_1d_lf4.py:26-30)
    uic = Expression('exp(-50*pow((x[0]-1), 2))')
    elastic.u0.assign(Function(elastic.U).interpolate(uic))
    sic = Expression('-exp(-50*pow((x[0]-1), 2))')
    elastic.s0.assign(Function(elastic.S).interpolate(sic))

    u1, s1 = elastic.run(T)
    return elastic, u1, s1
