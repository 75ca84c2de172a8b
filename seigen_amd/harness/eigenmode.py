"""Eigenmode verification problems, as ``tests/eigenmode/eigenmode_2d.py`` and
``tests/eigenmode/eigenmode_3d.py`` of the reference set them up."""
from math import pi, sqrt

from seigen_amd import (ElasticLF4, Expression, Function, UnitCubeMesh, UnitSquareMesh, Vp, Vs, log,
                        projected_abs_error_norm, timed_region)


class Eigenmode2DLF4():

    def __init__(self, N, degree, dt, solver='explicit', output=True, diagonal="left", quadrilateral=False):
        with timed_region('mesh generation'):
            # quadrilateral=True: the same script on UnitSquareMesh(N, N, quadrilateral=True) - create() is
            # family-agnostic (seigen/elastic.py:27-64, :81-82); "DG" then means the tensor-product element
            self.mesh = UnitSquareMesh(N, N, diagonal, quadrilateral=quadrilateral)

        self.elastic = ElasticLF4.create(self.mesh, "DG", degree, dimension=2,
                                         solver=solver, output=output)

        # Constants (eigenmode_2d.py:17-26)
        self.elastic.density = 1.0
        self.elastic.dt = dt
        self.elastic.mu = 0.25
        self.elastic.l = 0.5

        log("P-wave velocity: %f" % Vp(self.elastic.mu, self.elastic.l, self.elastic.density))
        log("S-wave velocity: %f" % Vs(self.elastic.mu, self.elastic.density))

        self.a = sqrt(2)*pi*Vs(self.elastic.mu, self.elastic.density)
        self.b = 2*pi*self.elastic.mu

    def _u(self, t):
        return Expression(('a*cos(pi*x[0])*sin(pi*x[1])*cos(a*t)',
                           '-a*sin(pi*x[0])*cos(pi*x[1])*cos(a*t)'), a=self.a, t=t)

    def _s(self, t):
        return Expression((('-b*sin(pi*x[0])*sin(pi*x[1])*sin(a*t)', '0'),
                           ('0', 'b*sin(pi*x[0])*sin(pi*x[1])*sin(a*t)')),
                          a=self.a, b=self.b, t=t)

    def eigenmode2d(self, T=5.0):
        # Initial conditions (eigenmode_2d.py:30-36): u at t=0, s at t=dt/2
        self.elastic.u0.assign(Function(self.elastic.U).interpolate(self._u(0)))
        self.elastic.s0.assign(Function(self.elastic.S).interpolate(self._s(self.elastic.dt/2.0)))
        return self.elastic.run(T)

    def eigenmode_error(self, u1, s1):
        # eigenmode_2d.py:40-65: exact fields at the hard-coded t=5 (stress at 5+dt/2),
        # |error| projected into DG6, L2 norm of the projection
        uexact = Function(self.elastic.U).interpolate(self._u(5))
        sexact = Function(self.elastic.S).interpolate(self._s(5 + self.elastic.dt/2.0))
        u_error = projected_abs_error_norm(u1, uexact, 6)
        s_error = projected_abs_error_norm(s1, sexact, 6)
        return u_error, s_error


class Eigenmode3DLF4():

    def __init__(self, N, degree, dt, solver='explicit', output=True, hexahedral=False):
        with timed_region('mesh generation'):
            # hexahedral=True: the same script on cubes - create() is family-agnostic (seigen/elastic.py:81-82)
            self.mesh = UnitCubeMesh(N, N, N, hexahedral=hexahedral)

        self.elastic = ElasticLF4.create(self.mesh, "DG", degree, dimension=3,
                                         solver=solver, output=output)

        # Constants (eigenmode_3d.py:17-26)
        self.elastic.density = 1.0
        self.elastic.dt = dt
        self.elastic.mu = 0.25
        self.elastic.l = 0.5

        log("P-wave velocity: %f" % Vp(self.elastic.mu, self.elastic.l, self.elastic.density))
        log("S-wave velocity: %f" % Vs(self.elastic.mu, self.elastic.density))

        self.A = sqrt(2*self.elastic.density*self.elastic.mu)
        self.O = pi*sqrt(2*self.elastic.mu/self.elastic.density)

    def _u(self, t):
        return Expression(('cos(pi*x[0])*(sin(pi*x[1]) - sin(pi*x[2]))*cos(O*t)',
                           'cos(pi*x[1])*(sin(pi*x[2]) - sin(pi*x[0]))*cos(O*t)',
                           'cos(pi*x[2])*(sin(pi*x[0]) - sin(pi*x[1]))*cos(O*t)'), O=self.O, t=t)

    def _s(self, t):
        return Expression((('-A*sin(pi*x[0])*(sin(pi*x[1]) - sin(pi*x[2]))*sin(O*t)', '0', '0'),
                           ('0', '-A*sin(pi*x[1])*(sin(pi*x[2]) - sin(pi*x[0]))*sin(O*t)', '0'),
                           ('0', '0', '-A*sin(pi*x[2])*(sin(pi*x[0]) - sin(pi*x[1]))*sin(O*t)')),
                          A=self.A, O=self.O, t=t)

    def eigenmode3d(self, T=5.0):
        # Initial conditions (eigenmode_3d.py:30-40)
        self.elastic.u0.assign(Function(self.elastic.U).interpolate(self._u(0)))
        self.elastic.s0.assign(Function(self.elastic.S).interpolate(self._s(self.elastic.dt/2.0)))
        return self.elastic.run(T)

    def eigenmode_error(self, u1, s1):
        # eigenmode_3d.py:42-69: DG3 projection
        uexact = Function(self.elastic.U).interpolate(self._u(5))
        sexact = Function(self.elastic.S).interpolate(self._s(5 + self.elastic.dt/2.0))
        u_error = projected_abs_error_norm(u1, uexact, 3)
        s_error = projected_abs_error_norm(s1, sexact, 3)
        return u_error, s_error


def convergence_analysis_2d(degrees=range(1, 5), N=(4, 8, 16, 32), out=None):
    """The sweep of eigenmode_2d.py:68-84 (dt = 0.5*(1/n)/2^(d-1), T=5)."""
    rows = []
    for d in degrees:
        for n in N:
            dt = 0.5*(1.0/n)/(2.0**(d-1))
            em = Eigenmode2DLF4(n, d, dt, output=False)
            u1, s1 = em.eigenmode2d()
            u_error, s_error = em.eigenmode_error(u1, s1)
            rows.append((d, 1.0/n, dt, u_error, s_error))
            if out:
                out.write("%d\t%g\t%g\t%r\t%r\n" % rows[-1])
    return rows


def convergence_analysis_3d(degrees=range(1, 4), N=(2, 4, 8), out=None):
    """The sweep of eigenmode_3d.py:72-88."""
    rows = []
    for d in degrees:
        for n in N:
            dt = 0.5*(1.0/n)/(2.0**(d-1))
            em = Eigenmode3DLF4(n, d, dt, output=False)
            u1, s1 = em.eigenmode3d()
            u_error, s_error = em.eigenmode_error(u1, s1)
            rows.append((d, 1.0/n, dt, u_error, s_error))
            if out:
                out.write("%d\t%g\t%g\t%r\t%r\n" % rows[-1])
    return rows
