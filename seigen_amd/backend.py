"""Thin object wrapper around one `sg_handle` (one mesh block on one GPU)."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import SgConfig, SgInfo, SgCounters, check


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def comm_unique_id():
    """A fresh RCCL unique id (bytes) for sg_comm_init: made on one rank, handed to all ranks of the block grid."""
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    check(_lib.load().sg_comm_get_unique_id(buf, _lib.COMM_ID_BYTES))
    return bytes(buf.raw)


def comm_library():
    """(path, version) of the RCCL the native exchange is bound to (sg_comm_library / sg_comm_version); raises where
    there is none."""
    buf = C.create_string_buffer(1024)
    check(_lib.load().sg_comm_library(buf, len(buf)))
    v = C.c_int()
    check(_lib.load().sg_comm_version(C.byref(v)))
    return buf.value.decode(), int(v.value)


class HipBlock(object):
    def __init__(self, dim, degree, n, h, origin, diagonal="left", nbr_mask=0, device=0, stream=None, dtype="f64",
                 cube0=None):
        """origin: the block's low corner - or, with `cube0` (the block's first cube counted from there), the origin
        of the whole mesh: node coordinates are then bitwise those of the unpartitioned mesh (sg_config::cube0)."""
        self.lib = _lib.load()
        if dtype not in ("f64", "f32"):
            raise ValueError("dtype must be 'f64' or 'f32'")
        self.dtype = dtype
        cfg = SgConfig()
        cfg.dim, cfg.degree = dim, degree
        cfg.dtype = 1 if dtype == "f32" else 0
        for a in range(3):
            cfg.n[a] = int(n[a]) if a < dim else 1
            cfg.h[a] = float(h[a]) if a < dim else 1.0
            cfg.origin[a] = float(origin[a]) if a < dim else 0.0
            cfg.cube0[a] = int(cube0[a]) if (cube0 is not None and a < dim) else 0
        cfg.diagonal = {"left": 0, "right": 1, "quadrilateral": 2}[diagonal]   # 2: the squares are the cells
        cfg.nbr_mask = int(nbr_mask)
        cfg.device = int(device)
        cfg.stream = C.c_void_p(stream) if stream else None
        hp = C.c_void_p()
        rc = self.lib.sg_create(C.byref(cfg), C.byref(hp))
        if rc != 0:
            raise _lib.SeigenHipError("sg_create failed (%d): %s" %
                                      (rc, (self.lib.sg_last_error(None) or b"").decode()))
        self.h = hp
        info = SgInfo()
        check(self.lib.sg_get_info(self.h, C.byref(info)), self.h)
        self.dim, self.degree = dim, degree
        self.nd, self.nf, self.nfaces, self.ncls = info.nd, info.nf, info.nfaces, info.nclasses
        self.ncells = int(info.ncells)
        self.u_dofs, self.s_dofs = int(info.u_dofs), int(info.s_dofs)
        self.halo_faces = [int(x) for x in info.halo_faces]
        self.nbr_mask = int(nbr_mask)

    def stream_ptr(self):
        """hipStream_t (as an integer) the block launches on."""
        p = C.c_void_p()
        check(self.lib.sg_get_stream(self.h, C.byref(p)), self.h)
        return p.value or 0

    def stream2_ptr(self):
        """hipStream_t of the stream that SG_REGION_SECOND launches of a split stage run on, or 0."""
        p = C.c_void_p()
        check(self.lib.sg_get_second_stream(self.h, C.byref(p)), self.h)
        return p.value or 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.sg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- shapes ---------------------------------------------------------------------
    def field_shape(self, field):
        d = self.dim
        if field in (_lib.FIELD_S, _lib.FIELD_SH):
            return (self.ncells, self.nd, d, d)
        return (self.ncells, self.nd, d)

    def node_coords(self, degree=None):
        degree = self.degree if degree is None else int(degree)
        nq = {1: degree + 1, 2: (degree + 1) * (degree + 2) // 2,
              3: (degree + 1) * (degree + 2) * (degree + 3) // 6}[self.dim]
        if self.nfaces == 2 * self.dim and self.dim > 1:      # tensor-product cells
            nq = (degree + 1) ** self.dim
        out = np.empty((self.ncells, nq, self.dim))
        check(self.lib.sg_node_coords(self.h, degree, out.ctypes.data, out.nbytes), self.h)
        return out

    # ---- data -------------------------------------------------------------------------
    def set_field(self, field, arr):
        arr = _f64(arr).reshape(self.field_shape(field))
        check(self.lib.sg_set_field(self.h, field, arr.ctypes.data, arr.nbytes), self.h)

    def get_field(self, field):
        out = np.empty(self.field_shape(field))
        check(self.lib.sg_get_field(self.h, field, out.ctypes.data, out.nbytes), self.h)
        return out

    def set_field_range(self, field, cell0, arr):
        arr = _f64(arr)
        ncells = arr.shape[0]
        check(self.lib.sg_set_field_range(self.h, field, int(cell0), int(ncells), arr.ctypes.data, arr.nbytes), self.h)

    def get_field_range(self, field, cell0, ncells, out=None):
        """`out`: a C-contiguous float64 array to fill (re-used buffers skip the page faults of a fresh one)"""
        shape = (int(ncells),) + self.field_shape(field)[1:]
        if out is None:
            out = np.empty(shape)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array of shape %r" % (shape,))
        check(self.lib.sg_get_field_range(self.h, field, int(cell0), int(ncells), out.ctypes.data, out.nbytes), self.h)
        return out

    def set_params(self, density, dt, lam, mu):
        lam_a, mu_a = _f64(np.atleast_1d(lam)).ravel(), _f64(np.atleast_1d(mu)).ravel()
        per_cell = int(lam_a.size > 1 or mu_a.size > 1)
        if per_cell:
            lam_a = _f64(np.broadcast_to(lam_a, (self.ncells,)))
            mu_a = _f64(np.broadcast_to(mu_a, (self.ncells,)))
        check(self.lib.sg_set_params(self.h, float(density), float(dt), lam_a.ctypes.data, mu_a.ctypes.data,
                                     per_cell), self.h)

    def set_density(self, rho, physical=False):
        """Scalar or per-cell density; physical=False: the explicit reference's u1 = rho*u0 + ...,
        True: u1 = u0 + (...)/rho (include/seigen_hip.h, sg_set_density)."""
        rho_a = _f64(np.atleast_1d(rho)).ravel()
        per_cell = int(rho_a.size > 1)
        if per_cell and rho_a.size != self.ncells:
            raise ValueError("per-cell density needs one value per cell of the block (%d), got %d" % (self.ncells, rho_a.size))
        check(self.lib.sg_set_density(self.h, rho_a.ctypes.data, per_cell, int(bool(physical))), self.h)

    def is_sym(self):
        v = C.c_int()
        check(self.lib.sg_get_sym(self.h, C.byref(v)), self.h)
        return bool(v.value)

    def leave_sym(self):
        check(self.lib.sg_leave_sym(self.h), self.h)

    def set_absorption(self, sigma_nodes, sigma_degree):
        if sigma_nodes is None:
            check(self.lib.sg_set_absorption(self.h, None, 0), self.h)
            return
        s = _f64(sigma_nodes).reshape(self.ncells, -1)
        check(self.lib.sg_set_absorption(self.h, s.ctypes.data, int(sigma_degree)), self.h)

    def set_source(self, nodes, values, static=False):
        """nodes: flat scalar node indices [nnz]; values [nsteps, nnz, d, d] for the next nsteps steps,
        or (static=True) [1, nnz, d, d] holding at every step."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int64).ravel()
        if nodes.size == 0 or values is None or len(values) == 0:
            check(self.lib.sg_set_source(self.h, 0, None, 0, None), self.h)
            return
        values = _f64(values).reshape(-1, nodes.size, self.dim, self.dim)
        if static and values.shape[0] != 1:
            raise ValueError("a static source is one time slice")
        check(self.lib.sg_set_source(self.h, nodes.size, nodes.ctypes.data, -1 if static else values.shape[0],
                                     values.ctypes.data), self.h)

    def set_source_box_ricker(self, lo, hi, a, t0, t_first, dt_step, nsteps):
        """The reference's box-Ricker source from its parameters (sg_set_source_box_ricker, include/seigen_hip.h)."""
        lo, hi = _f64(lo).ravel(), _f64(hi).ravel()
        if lo.size != self.dim or hi.size != self.dim:
            raise ValueError("lo and hi hold one coordinate per dimension")
        check(self.lib.sg_set_source_box_ricker(self.h, lo.ctypes.data, hi.ctypes.data, float(a), float(t0),
                                                float(t_first), float(dt_step), int(nsteps)), self.h)

    def set_source_separable(self, nodes, pattern, weights):
        """S(node, step k) = weights[k] * pattern[node]: nodes [nnz], pattern [nnz, d, d], weights [nsteps]."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int64).ravel()
        weights = _f64(weights).ravel()
        if nodes.size == 0 or weights.size == 0:
            check(self.lib.sg_set_source(self.h, 0, None, 0, None), self.h)
            return
        pattern = _f64(pattern).reshape(nodes.size, self.dim, self.dim)
        check(self.lib.sg_set_source_separable(self.h, nodes.size, nodes.ctypes.data, pattern.ctypes.data, weights.size,
                                               weights.ctypes.data), self.h)

    # ---- hot path ---------------------------------------------------------------------
    def step(self, nsteps=1):
        check(self.lib.sg_step(self.h, int(nsteps)), self.h)

    def run_stage(self, stage, region=_lib.REGION_ALL):
        check(self.lib.sg_run_stage(self.h, int(stage), int(region)), self.h)

    def end_step(self):
        check(self.lib.sg_end_step(self.h), self.h)

    def apply_F(self, s_in, u_abs, u_out):
        check(self.lib.sg_apply_F(self.h, s_in, u_abs, u_out), self.h)

    def apply_G(self, u_in, s_out, use_source=False):
        check(self.lib.sg_apply_G(self.h, u_in, s_out, int(use_source)), self.h)

    def sync(self):
        check(self.lib.sg_sync(self.h), self.h)

    def last_step_ms(self):
        ms = C.c_double()
        check(self.lib.sg_last_step_ms(self.h, C.byref(ms)), self.h)
        return ms.value

    def enable_timing(self, on=True):
        check(self.lib.sg_enable_timing(self.h, int(on)), self.h)

    def counters(self):
        c = SgCounters()
        check(self.lib.sg_get_counters(self.h, C.byref(c)), self.h)
        return dict(kernel_ms=list(c.kernel_ms), launches=list(c.launches), steps=int(c.steps),
                    halo_pack_ms=float(c.halo_pack_ms), halo_pack_launches=int(c.halo_pack_launches),
                    halo_bytes_packed=int(c.halo_bytes_packed))

    def stage_kernel_name(self, stage, region=0, short=True):
        """The kernel a launch of `stage` over `region` runs, named by the library itself (sg_stage_kernel_name): as
        rocprofv3 prints it, or (short) without the `void ` in front and the argument list behind."""
        buf = C.create_string_buffer(512)
        check(self.lib.sg_stage_kernel_name(self.h, int(stage), int(region), buf, len(buf)), self.h)
        name = buf.value.decode()
        if short:
            if name.startswith("void "):
                name = name[5:]
            if name.endswith(")") and "(" in name:
                name = name[:name.rindex("(")]
        return name

    # ---- halo ---------------------------------------------------------------------------
    def halo_bytes(self, field, side):
        nb = C.c_size_t()
        check(self.lib.sg_halo_bytes(self.h, field, side, C.byref(nb)), self.h)
        return nb.value

    def halo_pack(self, field, side, dev_ptr):
        check(self.lib.sg_halo_pack(self.h, field, side, C.c_void_p(dev_ptr)), self.h)

    def halo_pack_sides(self, field, ptrs):
        """ptrs: {side: device pointer}; every listed side in one launch."""
        arr = (C.c_void_p * 6)(*[C.c_void_p(ptrs.get(s)) if ptrs.get(s) else None for s in range(6)])
        check(self.lib.sg_halo_pack_sides(self.h, int(field), arr), self.h)

    # ---- native exchange over RCCL (csrc/comm.cpp): step() then runs the whole pipelined schedule in the library
    def comm_init(self, unique_id, rank, nranks, peers):
        """peers: rank of the face neighbour per block side (2 * axis + hi), -1 / None where there is none."""
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        _lib.check(self.lib.sg_comm_init(self.h, buf, _lib.COMM_ID_BYTES, int(rank), int(nranks), self._peer_array(peers)), self.h)

    def _peer_array(self, peers):
        return (C.c_int32 * 6)(*[(-1 if (i >= len(peers) or peers[i] is None) else int(peers[i])) for i in range(6)])

    def comm_check(self, rank, nranks, peers):
        """what sg_comm_init would refuse without talking to another rank (raises); call on every rank and agree on the
        outcome before the collective comm_init"""
        _lib.check(self.lib.sg_comm_check(self.h, int(rank), int(nranks), self._peer_array(peers)), self.h)

    def comm_selftest(self):
        """collective: pattern exchange; number of values that did not arrive from the facing side of the right neighbour"""
        bad = C.c_int64()
        _lib.check(self.lib.sg_comm_selftest(self.h, C.byref(bad)), self.h)
        return int(bad.value)

    def comm_finalize(self):
        _lib.check(self.lib.sg_comm_finalize(self.h), self.h)

    def comm_stats(self, reset=False):
        st = _lib.SgCommStats()
        _lib.check(self.lib.sg_comm_get_stats(self.h, C.byref(st), 1 if reset else 0), self.h)
        return {"exchanges": int(st.exchanges), "bytes_sent": int(st.bytes_sent), "exposed_wait_ms": float(st.exposed_wait_ms)}

    def comm_exchange(self, field):
        _lib.check(self.lib.sg_comm_exchange(self.h, field), self.h)

    def comm_buffers(self, kind, side):
        """(send pointer, receive pointer, bytes) of a side's device buffers; kind 0: velocity-like, 1: stress-like."""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        _lib.check(self.lib.sg_comm_buffers(self.h, kind, side, C.byref(a), C.byref(b), C.byref(n)), self.h)
        return a.value, b.value, n.value

    def halo_attach(self, field, side, dev_ptr):
        check(self.lib.sg_halo_attach(self.h, field, side, C.c_void_p(dev_ptr)), self.h)
