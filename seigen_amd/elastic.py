"""`ElasticLF4` - the solver-class API of ``seigen/elastic.py`` on MI355X.

Drop-in for the explicit path of the reference: same factory
(``ElasticLF4.create``, ``seigen/elastic.py:27-64``), same plain attributes
(``density dt mu l``), same ``absorption`` / ``source`` properties whose setters
interpolate an ``Expression`` into a pre-assigned ``Function`` (``:126-154``),
same ten named fields (``:93-103``) and the same ``run(T)`` loop (``:267-315``).
What Firedrake/PyOP2 generated and ran on CPU per timestep - eight assembles,
eight inverse-mass mat-vecs, two copies, halo exchanges - is six fused HIP
launches in libseigen_hip.so, driven through ctypes (include/seigen_hip.h).

Differences, all documented in DESIGN.md:
  * ``uh2`` and ``sh2`` are never materialised (fused into the combine stages);
    ``u0``/``u1`` (``s0``/``s1``) share one device buffer because
    ``u0.assign(u1)`` (``:296``) is an in-place update.
  * ``utemp`` has one consumer, ``sh2 = g(utemp)`` in the stress update ``s1 = s0 + dt sh1 + dt^3/24 sh2`` (``:302-303``),
    and ``g`` is linear: the library leaves ``w = dt u1 + dt^3/24 utemp`` in the buffer behind ``uh1`` / ``utemp`` and
    computes ``s1 = s0 + g(w)`` - one application of ``g`` that reads neither ``sh1`` nor a second right-hand side
    (include/seigen_hip.h, ``enum sg_stage``).  After a step ``elastic.utemp`` therefore holds ``w``; ``(u1, s1)`` are
    unchanged to round-off.
  * ``solver='implicit'`` (``:318-332``): the reference hands every form to a PETSc KSP; all eight
    systems are DG mass matrices (block diagonal), so the solve IS the element-wise inverse the
    explicit path applies, up to the KSP tolerance.  Here it runs the same six launches with the
    implicit forms' density convention (``form_u1``, ``:175-178``: u1 = u0 + (...)/rho).
  * VTK output (``:221-232``): ``velocity_<k>.vtu`` / ``stress_<k>.vtu`` streams with ``.pvd`` indices
    (``seigen_amd/vtu.py``; ``SEIGEN_OUTPUT=npy`` writes raw arrays instead).
  * The halo exchange implicit in every assemble (``:364``, ``:404-436``) runs inside the library over RCCL
    (``NativeExchanger``), or stage by stage from ``seigen_amd/parallel.py`` for process groups without device transport.
"""
import os
import sys
from contextlib import contextmanager

import numpy as np

from . import _lib
from .backend import HipBlock
from .functionspace import Function, FunctionSpace, TensorFunctionSpace, VectorFunctionSpace
from .helpers import log, allreduce_sum
from .parallel import world, HaloExchanger, NativeExchanger
from .profiling import timed_region

_SOLVER_MODES = ("implicit", "explicit", "parloop", "fusion", "tiling", "hip")


def _device_for_rank():
    if "SEIGEN_HIP_DEVICE" in os.environ:
        return int(os.environ["SEIGEN_HIP_DEVICE"])
    if world()[1] > 1:
        return int(os.environ.get("LOCAL_RANK", "0"))
    return 0


class ElasticLF4(object):
    r"""Elastic wave equation, DG in space, fourth-order leap-frog in time.
    Create with ``ElasticLF4.create(mesh, family, degree, dimension, solver)``."""

    @staticmethod
    def create(mesh, family, degree, dimension, solver="explicit", output=True, dtype="f64"):
        """Same signature and solver strings as ``seigen/elastic.py:27-64``; every
        explicit mode ('explicit', 'parloop', 'fusion', 'tiling', and the alias
        'hip') runs the fused HIP path.  ``dtype`` (not in the reference, which is double
        throughout, ``:442``): 'f64', or 'f32' for the FP32 second mode (3-D blocks)."""
        if solver == "implicit":
            return ImplicitElasticLF4(mesh, family, degree, dimension, output=output, dtype=dtype)
        elif solver in ("explicit", "hip"):
            return ExplicitElasticLF4(mesh, family, degree, dimension, output=output, dtype=dtype)
        elif solver == "parloop":
            return TilingElasticLF4(mesh, family, degree, dimension, output=output, dtype=dtype, tiling_mode=None)
        elif solver == "fusion":
            return TilingElasticLF4(mesh, family, degree, dimension, output=output, dtype=dtype, tiling_mode="hard")
        elif solver == "tiling":
            return TilingElasticLF4(mesh, family, degree, dimension, output=output, dtype=dtype, tiling_mode="tile")
        else:
            raise ValueError("Unknown solver mode. Must be one of: implicit, explicit, parloop")

    def __init__(self, mesh, family, degree, dimension, output=True, dtype="f64"):
        self.dtype = dtype
        with timed_region('function setup'):
            if dimension != mesh.dim:
                raise ValueError("dimension=%r does not match the mesh (%d-D)" % (dimension, mesh.dim))
            if not (1 <= int(degree) <= 4):
                raise ValueError("degree must be 1..4")
            self.mesh = mesh
            self.dimension = dimension
            self.degree = int(degree)
            self.output = output

            self.S = TensorFunctionSpace(mesh, family, degree, name='S')
            self.U = VectorFunctionSpace(mesh, family, degree, name='U')

            # Assumes that the S and U function spaces are the same.
            dofs = allreduce_sum(self.S.dof_count)
            log("Number of degrees of freedom: %d" % dofs)

            self._block = self._create_block()

            def bound(space, name, field):
                f = Function(space, name=name)
                if field is not None:
                    f._bind(self._block, field)
                return f

            self.s0 = bound(self.S, "StressOld", _lib.FIELD_S)
            self.sh1 = bound(self.S, "StressHalf1", _lib.FIELD_SH)
            self.stemp = bound(self.S, "StressTemp", _lib.FIELD_SH)
            self.sh2 = bound(self.S, "StressHalf2", None)       # fused away, never materialised
            self.s1 = bound(self.S, "StressNew", _lib.FIELD_S)

            self.u0 = bound(self.U, "VelocityOld", _lib.FIELD_U)
            self.uh1 = bound(self.U, "VelocityHalf1", _lib.FIELD_UH)
            self.utemp = bound(self.U, "VelocityTemp", _lib.FIELD_UH)
            self.uh2 = bound(self.U, "VelocityHalf2", None)      # fused away
            self.u1 = bound(self.U, "VelocityNew", _lib.FIELD_U)

            self.absorption_function = None
            self.source_function = None
            self.source_expression = None
            # Separable source S(x, t) = w(t) * source_function(x): when set (a callable t -> float),
            # `source_function` holds the spatial pattern and is NOT re-interpolated from
            # `source_expression`.  Build-defined (the reference only knows re-interpolated Expressions,
            # elastic.py:285-288): lets a harness use a pattern that is not a nodal interpolant, e.g.
            # the L2 projection of the source box (harness/explosive_source.py, source_mode).
            self.source_time_function = None
            self.density = None
            # False: the explicit reference's update u1 = rho*u0 + dt*uh1 + dt^3/24*uh2 (only rhs(form_u1)
            # is kept, elastic.py:341-345, :354-356); True: u1 = u0 + (...)/rho, what the implicit form
            # solves (elastic.py:175-178).  Identical for rho = 1, which every reference test uses.
            self.density_physical = False
            self.dt = None
            self.mu = None
            self.l = None

            self.invmass_velocity = None
            self.invmass_stress = None
            self._exchanger = None
            self._step_index = 0

        if self.output:
            with timed_region('i/o'):
                # File("velocity.pvd") / File("stress.pvd") of the reference (elastic.py:117-124);
                # SEIGEN_OUTPUT=npy writes raw arrays instead
                from .vtu import VtuStream
                rank, nranks = world()
                self.u_stream = VtuStream("velocity", rank, nranks)
                self.s_stream = VtuStream("stress", rank, nranks)

    # ---- device block ---------------------------------------------------------------------
    def _create_block(self):
        part = self.mesh.partition
        # the mesh's origin + the block's integer offset: coordinates bitwise those of the unpartitioned mesh
        block = HipBlock(self.mesh.dim, self.degree, part.n, self.mesh.h, list(self.mesh.origin),
                         "quadrilateral" if self.mesh.quadrilateral else self.mesh.diagonal,
                         part.nbr_mask, device=_device_for_rank(), dtype=self.dtype, cube0=list(part.start))
        self._torch_stream = None
        if part.world > 1:
            # One process per GPU: the library's launch stream, wrapped for torch, is made current
            # around every torch.distributed call (seigen_amd/parallel.py), so packs, exchanges and
            # the boundary launches are ordered on it.
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(_device_for_rank())
                self._torch_stream = torch.cuda.ExternalStream(block.stream_ptr(), device=_device_for_rank())
        return block

    @property
    def block(self):
        return self._block

    # ---- absorption / source (elastic.py:126-154) -----------------------------------------------
    @property
    def absorption(self):
        r"""The absorption coefficient :math:`\sigma` of the term :math:`\sigma\mathbf{u}`."""
        return self.absorption_function

    @absorption.setter
    def absorption(self, expression):
        self.absorption_function.interpolate(expression)

    @property
    def source(self):
        r"""The source term on the RHS of the stress equation."""
        return self.source_function

    @source.setter
    def source(self, expression):
        self.source_function.interpolate(expression)

    # ---- stage descriptors in place of the UFL forms (elastic.py:156-202) ------------------------
    form_uh1 = property(lambda self: ("uh1", _lib.STAGE_UH1))
    form_stemp = property(lambda self: ("stemp", _lib.STAGE_STEMP))
    form_uh2 = property(lambda self: ("uh2+u1", _lib.STAGE_U1))
    form_u1 = property(lambda self: ("uh2+u1", _lib.STAGE_U1))
    form_sh1 = property(lambda self: ("sh1", _lib.STAGE_SH1))
    form_utemp = property(lambda self: ("utemp", _lib.STAGE_UTEMP))
    form_sh2 = property(lambda self: ("sh2+s1", _lib.STAGE_S1))
    form_s1 = property(lambda self: ("sh2+s1", _lib.STAGE_S1))

    def write(self, u=None, s=None):
        r"""Write the velocity and/or stress fields (``seigen/elastic.py:221-232``): numbered
        ``.vtu`` files with point data named like the function (``VelocityNew``, ``StressNew``) and
        a ``.pvd`` index, as the reference's VTK streams; raw ``.npy`` with ``SEIGEN_OUTPUT=npy``."""
        if self.output:
            with timed_region('i/o'):
                if os.environ.get("SEIGEN_OUTPUT", "vtu") == "npy":
                    rank = world()[0]
                    if u:
                        np.save("velocity_%d_r%d.npy" % (self._step_index, rank), u.dat.data)
                    if s:
                        np.save("stress_%d_r%d.npy" % (self._step_index, rank), s.dat.data)
                    return
                if u:
                    self.u_stream.write(u, self._step_index * (self.dt or 0.0))
                if s:
                    self.s_stream.write(s, self._step_index * (self.dt or 0.0))

    def create_solver(self, form, result=None):
        """Solver context of one stage = its fused-launch id (``elastic.py:354-356``)."""
        return form[1]

    def solve(self, ctx, matrix=None, result=None):
        """Run one fused stage on the device (``elastic.py:358-367``).  The combine
        stages are fused into ``uh2``/``sh2``, so their contexts launch nothing new."""
        self._block.run_stage(ctx)

    def setup(self):
        """Upload parameters, sponge and source; stage contexts (``elastic.py:244-255``,
        ``:369-385`` - the inverse mass is folded into the reference operators)."""
        log("Creating solver contexts")
        with timed_region('solver setup'):
            for name in ("density", "dt", "mu", "l"):
                if getattr(self, name) is None:
                    raise ValueError("ElasticLF4.%s must be set before run()" % name)
            # density, l, mu: floats as in the reference's tests, or one value per cell of this
            # rank's block (build-defined heterogeneous extension)
            rho = np.atleast_1d(np.asarray(self.density, dtype=np.float64)).ravel()
            self._block.set_params(float(rho[0]), self.dt, self.l, self.mu)
            if rho.size > 1 or self.density_physical:
                self._block.set_density(rho if rho.size > 1 else float(rho[0]), self.density_physical)
            if self.absorption_function is not None:
                self._block.set_absorption(self.absorption_function.dat.data_cells,
                                           self.absorption_function.function_space().degree)
            else:
                self._block.set_absorption(None, 0)
            self.ctx_uh1 = self.create_solver(self.form_uh1, self.uh1)
            self.ctx_stemp = self.create_solver(self.form_stemp, self.stemp)
            self.ctx_uh2 = self.create_solver(self.form_uh2, self.uh2)
            self.ctx_u1 = self.create_solver(self.form_u1, self.u1)
            self.ctx_sh1 = self.create_solver(self.form_sh1, self.sh1)
            self.ctx_utemp = self.create_solver(self.form_utemp, self.utemp)
            self.ctx_sh2 = self.create_solver(self.form_sh2, self.sh2)
            self.ctx_s1 = self.create_solver(self.form_s1, self.s1)
            if self.mesh.partition.world > 1 and self._exchanger is None:
                import torch
                dev = torch.device("cuda", _device_for_rank())
                import torch.distributed as dist
                # SEIGEN_HALO_NATIVE: "1" (default) = inside the library when the process group is RCCL's; "0" = never;
                # "force" = also under a process group without device transport (gloo, MPI): the library makes its
                # own RCCL communicator, the group only carries the unique id and the ranks' agreement
                want = os.environ.get("SEIGEN_HALO_NATIVE", "1")
                native = (want == "force" or (dist.get_backend() == "nccl" and want != "0")) \
                    and os.environ.get("SEIGEN_HALO_SCHEDULE", "pipelined") != "plain"
                if native:      # the exchange inside the library: one C-ABI call per run of steps (csrc/comm.cpp)
                    # every rank must end up on the same path: agree on whether the communicator came up everywhere
                    try:
                        ex, failed = NativeExchanger(self._block, self.mesh.partition), 0
                    except Exception as e:      # noqa: BLE001 - whatever it was, the host-driven exchanger still works
                        ex, failed = None, 1
                        log("native halo exchange unavailable on this rank (%r): falling back to the host-driven one" % (e,))
                    if allreduce_sum(failed) > 0:
                        if ex is not None:
                            self._block.comm_finalize()
                        native = False
                    else:
                        self._exchanger = ex
                if not native:  # driven from here stage by stage (gloo / host-staged transports, the plain schedule)
                    self._exchanger = HaloExchanger(self._block, self.mesh.partition, dev, stream=self._torch_stream)

    @property
    def loop_context(self):
        @contextmanager
        def empty_loop_context():
            yield
        return empty_loop_context

    # ---- source handling -----------------------------------------------------------------------
    # a support this large means the source is not the localised kind the sparse table is meant for
    SOURCE_TABLE_MAX_BYTES = 2 << 30

    def _source_table(self, times):
        """Sparse source values: (nodes, values[nsteps or 1, nnz, d, d], static).

        The reference re-interpolates ``source_expression`` into ``source_function``
        at every step (``elastic.py:285-288``).  The same nodal values are computed
        here for all steps up front, but only on the support of the source, and uploaded
        once.  The support is found with ``t`` treated as unknown
        (``Expression.support_mask``): every node where the expression can be non-zero at
        ANY time - a time-windowed source is not missed, a source whose position depends on
        ``t`` gets the whole region it can reach.  A source without ``t`` is one time slice."""
        expr = self.source_expression
        nsteps = len(times)
        d = self.dimension
        if self.source_time_function is not None:
            vals = self.source_function.dat.data_cells
            nz = np.nonzero(np.abs(vals).reshape(vals.shape[0] * vals.shape[1], -1).max(axis=1) > 0)[0]
            v = vals.reshape(-1, d, d)[nz]
            w = np.array([float(self.source_time_function(t)) for t in times])
            return nz, w[:, None, None, None] * v[None], False
        if expr is None or not hasattr(expr, "_params") or "t" not in expr._params:
            vals = self.source_function.dat.data_cells
            nz = np.nonzero(np.abs(vals).reshape(vals.shape[0] * vals.shape[1], -1).max(axis=1) > 0)[0]
            v = vals.reshape(-1, d, d)[nz]
            return nz, v[None], True
        t_keep = expr.t
        nz, Xs = [], []
        for cell0, X, cells in self._support_scan_chunks(expr):
            idx = np.nonzero(expr.support_mask(X).reshape(-1))[0]
            if cells is None:
                nz.append(idx + cell0 * X.shape[1])
            else:                                         # a sub-box of the block: rows of X are the cells `cells`
                nz.append(cells[idx // X.shape[1]] * X.shape[1] + idx % X.shape[1])
            Xs.append(X.reshape(-1, X.shape[-1])[idx])
        nz, Xs = np.concatenate(nz), np.concatenate(Xs)
        if nsteps * len(nz) * d * d * 8 > self.SOURCE_TABLE_MAX_BYTES:
            raise MemoryError("the source can be non-zero at %d nodes over %d steps: its table would take %.1f GB; "
                              "too much for a per-step table, and the source does not factorise into w(t) * pattern(x)"
                              % (len(nz), nsteps, nsteps * len(nz) * d * d * 8 / 1e9))
        values = expr.evaluate_times(Xs, times).reshape(nsteps, len(nz), d, d)
        expr.t = t_keep
        return nz, values, False

    def _support_scan_chunks(self, expr):
        """Node coordinates to search for the support of a source: the whole block slab by slab (bounded host
        memory), or - when the expression carries a hint `support_box = (lo, hi)`, the caller's promise that it
        vanishes outside that box - only the cubes of this rank's block that touch the box (a 128^3-cube P4
        block has 440 M nodes; a localised source touches a few thousand).  Yields (cell0, X, cells): X
        [n, nd, dim]; cells = None for n consecutive cells from cell0, else the block-local cell of every row."""
        box = getattr(expr, "support_box", None)
        if box is None:
            for cell0, X in self.S.node_coords_chunks():
                yield cell0, X, None
            return
        import ctypes as C
        from .functionspace import block_config
        mesh, part, d = self.mesh, self.mesh.partition, self.dimension
        ncls = mesh.cells_per_block
        lo, hi = np.asarray(box[0], dtype=np.float64), np.asarray(box[1], dtype=np.float64)
        rng = []
        for a in range(d):
            o = mesh.origin[a] + part.start[a] * mesh.h[a]
            i0 = max(int(np.floor((lo[a] - o) / mesh.h[a] - 1e-9)), 0)
            i1 = min(int(np.floor((hi[a] - o) / mesh.h[a] + 1e-9)), part.n[a] - 1)
            if i1 < i0:
                return
            rng.append((i0, i1 - i0 + 1))
        cfg = block_config(mesh, self.degree)
        for a in range(d):
            cfg.n[a] = rng[a][1]
            cfg.cube0[a] = part.start[a] + rng[a][0]      # integer offset: the coordinates of the full scan, bit for bit
        nsub = int(np.prod([r[1] for r in rng])) * ncls
        X = np.empty((nsub, self.S.nd, d))
        _lib.check(_lib.load().sg_block_node_coords(C.byref(cfg), self.degree, X.ctypes.data, X.nbytes))
        # sub-box cube (x fastest) -> block cube -> cell = cube * ncls + class
        idx = np.indices([r[1] for r in reversed(rng)]).reshape(d, -1)[::-1]        # [axis][sub cube], x fastest
        cube = np.zeros(idx.shape[1], dtype=np.int64)
        mul = 1
        for a in range(d):
            cube += (idx[a] + rng[a][0]) * mul
            mul *= part.n[a]
        cells = (cube[:, None] * ncls + np.arange(ncls)[None, :]).reshape(-1)
        yield 0, X, cells

    def _source_separable(self, times):
        """(nodes, pattern [nnz, d, d], weights [nsteps]) if the source is S(x, t) = w(t) * pattern(x) on its
        support - what `cond(x) ? f(t) : 0` sources (explosive_source_lf4.py:36-40) and `source_time_function` are -
        else None.  An Expression is TESTED, not assumed: the factorisation found at one pivot entry must
        reproduce the full evaluation at six other steps to round-off."""
        d = self.dimension
        if self.source_time_function is not None:
            vals = self.source_function.dat.data_cells
            nz = np.nonzero(np.abs(vals).reshape(vals.shape[0] * vals.shape[1], -1).max(axis=1) > 0)[0]
            return nz, vals.reshape(-1, d, d)[nz], np.array([float(self.source_time_function(t)) for t in times])
        expr = self.source_expression
        if expr is None or not hasattr(expr, "_params") or "t" not in expr._params or len(times) < 1:
            return None
        t_keep = expr.t
        try:
            nz, Xs = [], []
            for cell0, X, cells in self._support_scan_chunks(expr):
                idx = np.nonzero(expr.support_mask(X).reshape(-1))[0]
                nz.append(idx + cell0 * X.shape[1] if cells is None else cells[idx // X.shape[1]] * X.shape[1] + idx % X.shape[1])
                Xs.append(X.reshape(-1, X.shape[-1])[idx])
            nz, Xs = np.concatenate(nz), np.concatenate(Xs)
            if len(nz) == 0:
                return nz, np.zeros((0, d, d)), np.zeros(len(times))
            n = len(times)
            best = None
            for k in sorted({0, n // 4, n // 2, 3 * n // 4, n - 1}):          # a step where the source is strong
                expr.t = times[k]
                V = expr.evaluate(Xs).reshape(len(nz), d, d)
                if best is None or np.abs(V).max() > np.abs(best[1]).max():
                    best = (k, V)
            k0, pattern = best
            if not np.abs(pattern).max() > 0:
                return None
            piv = np.unravel_index(np.abs(pattern).argmax(), pattern.shape)
            xp = Xs[piv[0]:piv[0] + 1]
            w = expr.evaluate_times(xp, times).reshape(n, d, d)[:, piv[1], piv[2]] / pattern[piv]
            rng = np.random.default_rng(0)
            for k in sorted(set(int(v) for v in rng.integers(0, n, size=6)) | {0, n - 1}):
                expr.t = times[k]
                V = expr.evaluate(Xs).reshape(len(nz), d, d)
                if np.abs(V - w[k] * pattern).max() > 1e-13 * max(np.abs(V).max(), np.abs(w[k] * pattern).max(), 1e-300):
                    return None
            return nz, pattern, w
        finally:
            expr.t = t_keep

    def upload_source(self, times):
        """Hand the source of the next len(times) steps to the device: as a table of nodal values per step
        (the reference's re-interpolation, elastic.py:285-288, restricted to the support) or, when that table would
        exceed SOURCE_TABLE_MAX_BYTES or `source_time_function` is set, as one slice and a weight per step
        (sg_set_source_separable) if the source factorises."""
        if not self.source:
            self._block.set_source([], None)
            return
        if self.source_time_function is not None:
            self._block.set_source_separable(*self._source_separable(times))
            return
        try:
            nodes, values, static = self._source_table(times)
        except MemoryError:
            sep = self._source_separable(times)
            if sep is None:
                raise
            self._block.set_source_separable(*sep)
            return
        self._block.set_source(nodes, values, static=static)

    # ---- time loop (elastic.py:267-315) ----------------------------------------------------------
    def step_times(self, T):
        """The values of `t` visited by the reference loop ``t = dt; while t <= T + 1e-12``."""
        times = []
        t = self.dt
        while t <= T + 1e-12:
            times.append(t)
            t += self.dt
        return times

    def _advance(self, nsteps):
        if self._exchanger is not None:
            if os.environ.get("SEIGEN_HALO_SCHEDULE", "pipelined") == "plain":
                self._exchanger.step_unpipelined(nsteps)      # input traces || interior, then the shell
            else:
                self._exchanger.step(nsteps)
        else:
            self._block.step(nsteps)
        self._step_index += nsteps

    def _agree_on_stress_storage(self):
        """A block leaves symmetric-stress storage when IT is handed a non-symmetric stress or
        source; its neighbours read its traces, so all blocks of the mesh must store alike."""
        if self.mesh.partition.world > 1:
            mine = 0 if self._block.is_sym() else 1
            if allreduce_sum(mine) > 0 and not mine:
                self._block.leave_sym()

    def run(self, T):
        """Run the elastic wave simulation until t = T; returns (u1, s1)."""
        # Write out the initial condition.
        self.write(self.u1, self.s1)

        # Call solver-specific setup
        self.setup()

        with timed_region('timestepping'):
            times = self.step_times(T)
            with timed_region('source term update'):
                self.upload_source(times)
            self._agree_on_stress_storage()
            with self.loop_context():
                if self.output:
                    for t in times:
                        log("t = %f" % t)
                        self._advance(1)
                        self.write(self.u1, self.s1)
                else:
                    # the eight solves + two assigns of every step, without returning to Python
                    self._advance(len(times))
            self._block.sync()
            if self.source and self.source_expression is not None and times and \
                    self.source_time_function is None and \
                    "t" in getattr(self.source_expression, "_params", {}):
                self.source_expression.t = times[-1]
                self.source = self.source_expression
        return self.u1, self.s1


class ImplicitElasticLF4(ElasticLF4):
    r"""``solver='implicit'`` (``seigen/elastic.py:318-332``): the reference wraps each of the eight
    forms in a ``LinearVariationalSolver``.  Every left-hand side is a DG mass matrix - block diagonal,
    one dense block per cell - so what the KSP converges to is the element-wise inverse applied to the
    assembled right-hand side: the arithmetic of the explicit path, which the GPU kernels fuse into the
    stage launches (no iteration, no tolerance; the reference's answer differs from it by its KSP
    tolerance).  What does differ between the two reference classes is ``form_u1``: the implicit one
    keeps the density on the left (``:175-178``: rho (u - u0)/dt = uh1 + dt^2/24 uh2), the explicit one
    multiplies u0 by it (``:341-345``).  This class therefore runs the same six launches with
    ``density_physical = True``; for rho = 1 (every reference test) the two coincide."""

    def __init__(self, *args, **kwargs):
        super(ImplicitElasticLF4, self).__init__(*args, **kwargs)
        self.density_physical = True


class ExplicitElasticLF4(ElasticLF4):
    r"""Explicit solves: RHS assembly and element-wise inverse mass fused on the GPU
    (``seigen/elastic.py:335-385``)."""
    pass


class TilingElasticLF4(ExplicitElasticLF4):
    r"""'parloop' / 'fusion' / 'tiling' of the reference (``seigen/elastic.py:388-515``)
    differ from 'explicit' only in HOW the CPU executes the same arithmetic (per-cell
    mat-vec par_loop, PyOP2 loop fusion, SLOPE tiling).  On the GPU that role is played
    by kernel fusion inside libseigen_hip, so these modes share the explicit path."""

    loop_chain_length = 28
    num_solves = 8
    tile_size = 1000
    extra_halo = 0

    def __init__(self, mesh, *args, **kwargs):
        self.tiling_mode = kwargs.pop("tiling_mode", None)
        self.num_unroll = 0 if self.tiling_mode is None else 1
        super(TilingElasticLF4, self).__init__(mesh, *args, **kwargs)

    def calculate_sdepth(self, num_solves, num_unroll, extra_halo):
        """Halo depth the reference would request (``seigen/elastic.py:422-436``); the HIP
        path always exchanges depth-1 facet traces per stage."""
        if world()[1] > 1:
            return 1 + num_solves * num_unroll + extra_halo
        return 1
