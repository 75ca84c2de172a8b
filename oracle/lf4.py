"""Fourth-order leap-frog driver.  ORACLE (test infrastructure).

Restates ``ElasticLF4.run`` (``seigen/elastic.py:267-315``) with the explicit
solver's stage forms (``ExplicitElasticLF4.form_u1/form_s1``, ``:340-352``):

  t = dt; while t <= T + 1e-12:                         elastic.py:279-280
    source re-interpolated at time t                    elastic.py:285-288
    uh1   = Minv f(s0; u0)                               :292, :157-160
    stemp = Minv g(uh1)                                  :293, :163-166
    uh2   = Minv f(stemp; u0)                            :294, :169-172
    u1    = rho*u0 + dt*uh1 + dt^3/24*uh2                :295, :341-345 (rhs only, unweighted Minv :376)
    u0 <- u1                                             :296
    sh1   = Minv g(u1)                                   :300, :181-184
    utemp = Minv f(sh1; u1)                              :301, :187-190
    sh2   = Minv g(utemp)                                :302, :193-196
    s1    = s0 + dt*sh1 + dt^3/24*sh2                    :303, :348-352
    s0 <- s1                                             :304
    t += dt                                              :313
"""
import numpy as np
from .forms import ElasticOperators


def count_steps(dt, T):
    """Number of iterations of the reference's loop (repeated fp addition)."""
    n = 0
    t = dt
    while t <= T + 1e-12:
        n += 1
        t += dt
    return n


class OracleLF4(object):
    def __init__(self, mesh, degree):
        self.mesh = mesh
        self.degree = degree
        self.E = ElasticOperators(mesh, degree)
        d, nd, nc = mesh.dim, self.E.nd, mesh.ncells
        self.dim = d
        self.u0 = np.zeros((nc, nd, d))
        self.s0 = np.zeros((nc, nd, d, d))
        self.u1 = np.zeros((nc, nd, d))
        self.s1 = np.zeros((nc, nd, d, d))
        self.density = 1.0           # float or one value per cell
        self.density_physical = False  # False: rho*u0 + ... (explicit reference, :341-345); True: u0 + (...)/rho (:175-178)
        self.dt = None
        self.mu = None
        self.l = None
        self.source = None          # callable t -> [nc, nd, d, d]
        self.probes = None          # optional callback(step, t, u1, s1)

    def node_coords(self):
        return self.mesh.node_coords(self.degree)

    def step(self, t):
        E, dt, rho = self.E, self.dt, self.density
        S = self.source(t) if self.source is not None else None
        uh1 = E.apply_F(self.s0, self.u0)
        stemp = E.apply_G(uh1, self.l, self.mu, S)
        uh2 = E.apply_F(stemp, self.u0)
        if np.ndim(rho):
            rho = np.asarray(rho, dtype=np.float64).reshape(-1, 1, 1)
        if self.density_physical:
            self.u1 = self.u0 + (dt * uh1 + (dt ** 3 / 24.0) * uh2) / rho
        else:
            self.u1 = rho * self.u0 + dt * uh1 + (dt ** 3 / 24.0) * uh2
        self.u0 = self.u1
        sh1 = E.apply_G(self.u1, self.l, self.mu, S)
        utemp = E.apply_F(sh1, self.u1)
        sh2 = E.apply_G(utemp, self.l, self.mu, S)
        self.s1 = self.s0 + dt * sh1 + (dt ** 3 / 24.0) * sh2
        self.s0 = self.s1
        self.last = dict(uh1=uh1, stemp=stemp, uh2=uh2, sh1=sh1, utemp=utemp, sh2=sh2)

    def run(self, T, max_steps=None):
        t = self.dt
        n = 0
        while t <= T + 1e-12:
            self.step(t)
            n += 1
            if self.probes is not None:
                self.probes(n, t, self.u1, self.s1)
            if max_steps is not None and n >= max_steps:
                break
            t += self.dt
        self.nsteps = n
        return self.u1, self.s1
