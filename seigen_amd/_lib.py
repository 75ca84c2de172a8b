"""ctypes binding of libseigen_hip.so (C-ABI: include/seigen_hip.h).

The HIP library is the only compute back end.  If it is missing this module
raises - there is no CPU fallback to fall through to.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_ENV = "SEIGEN_HIP_LIB"

SG_OK = 0
ABI_VERSION = 2      # SG_ABI_VERSION of include/seigen_hip.h this binding was written against
FIELD_U, FIELD_UH, FIELD_S, FIELD_SH = 0, 1, 2, 3
STAGE_UH1, STAGE_STEMP, STAGE_U1, STAGE_SH1, STAGE_UTEMP, STAGE_S1 = range(6)
REGION_ALL, REGION_INTERIOR, REGION_BOUNDARY, REGION_FIRST, REGION_SECOND = 0, 1, 2, 3, 4


class SgConfig(C.Structure):
    _fields_ = [("dim", C.c_int32), ("degree", C.c_int32), ("n", C.c_int32 * 3),
                ("h", C.c_double * 3), ("origin", C.c_double * 3),
                ("diagonal", C.c_int32), ("nbr_mask", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32),
                ("stream", C.c_void_p), ("cube0", C.c_int32 * 3), ("pad_", C.c_int32)]


class SgInfo(C.Structure):
    _fields_ = [("dim", C.c_int32), ("degree", C.c_int32), ("nd", C.c_int32), ("nf", C.c_int32),
                ("nfaces", C.c_int32), ("nclasses", C.c_int32), ("ncells", C.c_int64),
                ("u_dofs", C.c_int64), ("s_dofs", C.c_int64), ("halo_faces", C.c_int32 * 6)]


class SgCommStats(C.Structure):
    _fields_ = [("exchanges", C.c_int64), ("bytes_sent", C.c_int64), ("exposed_wait_ms", C.c_double)]


COMM_ID_BYTES = 128


class SgCounters(C.Structure):
    _fields_ = [("kernel_ms", C.c_double * 6), ("launches", C.c_int64 * 6), ("steps", C.c_int64),
                ("halo_pack_ms", C.c_double), ("halo_pack_launches", C.c_int64), ("halo_bytes_packed", C.c_int64)]


# every symbol include/seigen_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_DP = C.POINTER(C.c_double)
SYMBOLS = {
    "sg_abi_version": (C.c_int, []),
    "sg_create": (C.c_int, [C.POINTER(SgConfig), C.POINTER(_P)]),
    "sg_destroy": (None, [_P]),
    "sg_last_error": (C.c_char_p, [_P]),
    "sg_get_info": (C.c_int, [_P, C.POINTER(SgInfo)]),
    "sg_sync": (C.c_int, [_P]),
    "sg_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "sg_get_second_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "sg_node_coords": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "sg_block_node_coords": (C.c_int, [C.POINTER(SgConfig), C.c_int, _P, C.c_size_t]),
    "sg_set_params": (C.c_int, [_P, C.c_double, C.c_double, _P, _P, C.c_int]),
    "sg_set_density": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "sg_set_field": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "sg_get_field": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "sg_set_field_range": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, C.c_size_t]),
    "sg_get_field_range": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, C.c_size_t]),
    "sg_set_absorption": (C.c_int, [_P, _P, C.c_int]),
    "sg_set_source": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P]),
    "sg_set_source_separable": (C.c_int, [_P, C.c_int64, _P, _P, C.c_int64, _P]),
    "sg_set_source_box_ricker": (C.c_int, [_P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64]),
    "sg_step": (C.c_int, [_P, C.c_int64]),
    "sg_run_stage": (C.c_int, [_P, C.c_int, C.c_int]),
    "sg_end_step": (C.c_int, [_P]),
    "sg_apply_F": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "sg_apply_G": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "sg_halo_bytes": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "sg_halo_pack": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "sg_halo_pack_sides": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "sg_halo_attach": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "sg_comm_get_unique_id": (C.c_int, [_P, C.c_size_t]),
    "sg_comm_init": (C.c_int, [_P, _P, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "sg_comm_version": (C.c_int, [C.POINTER(C.c_int)]),
    "sg_comm_library": (C.c_int, [C.c_char_p, C.c_size_t]),
    "sg_comm_check": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "sg_comm_selftest": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "sg_comm_finalize": (C.c_int, [_P]),
    "sg_comm_get_stats": (C.c_int, [_P, C.POINTER(SgCommStats), C.c_int]),
    "sg_comm_exchange": (C.c_int, [_P, C.c_int]),
    "sg_comm_buffers": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "sg_get_sym": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "sg_leave_sym": (C.c_int, [_P]),
    "sg_enable_timing": (C.c_int, [_P, C.c_int]),
    "sg_get_counters": (C.c_int, [_P, C.POINTER(SgCounters)]),
    "sg_last_step_ms": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "sg_reference_operator": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t]),
    "sg_stage_kernel_name": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "sg_region_boxes": (C.c_int, [C.POINTER(SgConfig), C.c_int, _P, C.c_int]),
    "sg_tabulate": (C.c_int, [C.c_int, C.c_int, C.c_int64, _P, _P]),
    "sg_tabulate_cell": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P]),
    "sg_reference_operator_cell": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_size_t]),
    "sg_mesh_tables": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
}

_lib = None


class SeigenHipError(RuntimeError):
    pass


def lib_path():
    return os.environ.get(LIB_ENV) or os.path.join(_HERE, "libseigen_hip.so")


def load():
    """Load the HIP library or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SeigenHipError(
            "libseigen_hip.so not found at %s - build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C seigen_amd/csrc`). "
            "seigen_amd has no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    have = lib.sg_abi_version()
    if have != ABI_VERSION:
        raise SeigenHipError("%s was built with SG_ABI_VERSION %d, this binding expects %d (include/seigen_hip.h): rebuild the "
                             "library" % (path, have, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != SG_OK:
        msg = load().sg_last_error(handle)
        raise SeigenHipError("libseigen_hip error %d: %s" % (rc, (msg or b"").decode()))
