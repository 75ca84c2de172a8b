"""The 1-D demonstration of the reference (``tests/pulse/pulse_1d_lf4.py``): a Gaussian velocity pulse with the
matching stress on a bar of length 4, DG1, absorbing layers half a unit wide at both ends, run to T = 2.

The reference's file is a flat script; here the set-up is a small class with its numbers as parameters (defaults =
the script's: ``:7-8`` mesh, ``:13-17`` constants, ``:22-24`` sponge, ``:26-30`` initial state, ``:32-33`` end time),
and ``pulse_1d_lf4`` is the one-call form the tests use."""
from seigen_amd import ElasticLF4, Expression, Function, FunctionSpace, IntervalMesh, Vp, Vs, log, timed_region


class Pulse1D(object):
    def __init__(self, length=4.0, cell=1e-2, degree=1, solver="explicit", output=False):
        self.length = float(length)
        with timed_region('mesh generation'):
            self.mesh = IntervalMesh(int(self.length / cell), self.length)
        self.elastic = ElasticLF4.create(self.mesh, "DG", degree, dimension=1, solver=solver, output=output)

    def material(self, density=1.0, mu=0.25, lam=0.5, dt=0.0025):
        el = self.elastic
        el.density, el.mu, el.l, el.dt = density, mu, lam, dt
        for name, speed in (("P", Vp(mu, lam, density)), ("S", Vs(mu, density))):
            log("%s-wave velocity: %f" % (name, speed))
        return self

    def absorbing_ends(self, width=0.5, sigma=100.0, space_degree=1):
        """sigma in the layers x <= width and x >= length - width, interpolated into a DG space of its own"""
        el = self.elastic
        el.absorption_function = Function(FunctionSpace(el.mesh, "DG", space_degree))
        el.absorption = Expression("x[0] >= %r || x[0] <= %r ? %r : 0" % (self.length - width, width, sigma))
        return self

    def gaussian_pulse(self, centre=1.0, sharpness=50.0):
        """u = exp(-sharpness (x - centre)^2), s = -u: a pulse travelling to the right"""
        el = self.elastic
        bump = 'exp(-%r*pow((x[0]-%r), 2))' % (sharpness, centre)
        el.u0.assign(Function(el.U).interpolate(Expression(bump)))
        el.s0.assign(Function(el.S).interpolate(Expression('-' + bump)))
        return self

    def run(self, T=2.0):
        return self.elastic.run(T)


def pulse_1d_lf4(T=2.0, Lx=4.0, h=1e-2, dt=0.0025, degree=1, solver="explicit", output=False):
    p = Pulse1D(Lx, h, degree, solver, output).material(dt=dt).absorbing_ends().gaussian_pulse()
    u1, s1 = p.run(T)
    return p.elastic, u1, s1
