"""Explosive source in a half space, as
``tests/explosive_source/explosive_source_lf4.py`` of the reference."""
from seigen_amd import (ElasticLF4, Expression, Function, FunctionSpace, RectangleMesh, Vp, Vs, cfl_dt, log,
                        timed_region)


class ExplosiveSourceLF4():

    def generate_mesh(self, Lx=300.0, Ly=150.0, h=2.5, quadrilateral=False):
        return RectangleMesh(int(Lx/h), int(Ly/h), Lx, Ly, quadrilateral=quadrilateral)

    def setup(self, Lx=300.0, Ly=150.0, h=2.5, degree=2, solver="explicit", output=False,
              courant_number=0.5, dt=None, source_x=45.0, source_mode="interpolate", source_y=None,
              quadrilateral=False, dtype="f64"):
        """source_mode: 'interpolate' - the reference's nodal interpolation of the box indicator
        (explosive_source_lf4.py:36-40; its integral depends on the mesh: 2.08 m^2 at h = 2.5, P2);
        'unit_integral' - the same interpolant scaled so that its integral is the box's 1 m^2;
        'project' - the L2 projection of the indicator (integral 1 m^2 on every mesh)."""
        with timed_region('mesh generation'):
            mesh = self.generate_mesh(Lx, Ly, h, quadrilateral)
            self.elastic = ElasticLF4.create(mesh, "DG", degree, dimension=2,
                                             solver=solver, output=output, dtype=dtype)

        # Constants (explosive_source_lf4.py:21-23)
        self.elastic.density = 1.0
        self.elastic.mu = 3600.0
        self.elastic.l = 3599.3664

        self.Vp = Vp(self.elastic.mu, self.elastic.l, self.elastic.density)
        self.Vs = Vs(self.elastic.mu, self.elastic.density)
        log("P-wave velocity: %f" % self.Vp)
        log("S-wave velocity: %f" % self.Vs)

        self.dx = h
        self.courant_number = courant_number
        # explosive_source_lf4.py:30-32 uses cfl_dt(dx, Vp, 0.5); that step is unstable for the
        # explicit sponge (sigma*dt = 12), so callers may pass the stable dt=0.001 of uy.py:25
        self.elastic.dt = cfl_dt(self.dx, self.Vp, self.courant_number) if dt is None else dt
        log("Using a timestep of %f" % self.elastic.dt)

        # Source (explosive_source_lf4.py:35-40): Ricker wavelet in a 1 m box, 1 m below the surface
        a = 159.42
        source_y = Ly - 1.0 if source_y is None else source_y
        box = "x[0] >= %r && x[0] <= %r && x[1] >= %r && x[1] <= %r" % (
            source_x - 0.5, source_x + 0.5, source_y - 0.5, source_y + 0.5)
        ricker = "(-1.0 + 2*a*pow(t - 0.3, 2))*exp(-a*pow(t - 0.3, 2))"
        code = "%s ? %s : 0.0" % (box, ricker)
        self.elastic.source_expression = Expression(((code, "0.0"), ("0.0", code)), a=a, t=0)
        self.elastic.source_function = Function(self.elastic.S)
        self.elastic.source = self.elastic.source_expression
        if source_mode != "interpolate":
            import math
            import numpy as np
            from seigen_amd.functionspace import integral, project_box_indicator
            el = self.elastic
            lo, hi = (source_x - 0.5, source_y - 0.5), (source_x + 0.5, source_y + 0.5)
            scalar = Function(FunctionSpace(mesh, "DG", degree))
            if source_mode == "project":
                scalar.assign(project_box_indicator(scalar.function_space(), lo, hi))
            elif source_mode == "unit_integral":
                scalar.interpolate(Expression("%s ? 1.0 : 0.0" % box))
                scalar.assign(scalar.dat.data_cells * ((hi[0] - lo[0]) * (hi[1] - lo[1]) / float(integral(scalar))))
            else:
                raise ValueError("source_mode must be 'interpolate', 'unit_integral' or 'project'")
            self.source_integral = float(integral(scalar))
            pattern = np.zeros(scalar.dat.data_cells.shape + (2, 2))
            pattern[..., 0, 0] = pattern[..., 1, 1] = scalar.dat.data_cells
            el.source_function.assign(pattern)
            el.source_time_function = lambda t: (-1.0 + 2 * a * (t - 0.3) ** 2) * math.exp(-a * (t - 0.3) ** 2)

        # Absorption (explosive_source_lf4.py:42-45): DG4 sponge, 20 m wide, not on the free surface
        F = FunctionSpace(mesh, "DG", 4)
        self.elastic.absorption_function = Function(F)
        self.elastic.absorption = Expression("x[0] <= 20 || x[0] >= %r || x[1] <= 20.0 ? 1000 : 0" % (Lx - 20.0))

        # Initial conditions (explosive_source_lf4.py:47-52)
        uic = Expression(('0.0', '0.0'))
        self.elastic.u0.assign(Function(self.elastic.U).interpolate(uic))
        sic = Expression((('0', '0'),
                          ('0', '0')))
        self.elastic.s0.assign(Function(self.elastic.S).interpolate(sic))
        return self.elastic

    def explosive_source_lf4(self, T=2.5, Lx=300.0, Ly=150.0, h=2.5, solver="explicit", output=False, **kw):
        self.setup(Lx, Ly, h, solver=solver, output=output, **kw)
        # Start the simulation
        with timed_region('elastic-run'):
            return self.elastic.run(T)

    def record_receivers(self, T, receivers=((45.0, 149.0), (90.0, 149.0), (140.0, 149.0)), every=5):
        """Run to T and sample VelocityNew at the receivers every `every` steps - the counterpart of
        tests/explosive_source/uy.py:31-43 (which probes the VTU files of every 5th step).
        Returns times [n], traces [n, nrecv, 2]."""
        import numpy as np
        from seigen_amd.functionspace import evaluate_at, locate
        el = self.elastic
        el.setup()
        times = el.step_times(T)
        el.upload_source(times)
        locs = [locate(el.U, r) for r in receivers]
        out_t, out_v = [], []
        done = 0
        while done + every <= len(times):
            el._advance(every)
            done += every
            out_t.append(times[done - 1])
            out_v.append([evaluate_at(el.u1, r, loc) for r, loc in zip(receivers, locs)])
        return np.array(out_t), np.array(out_v)
