"""ctypes binding of libemdee_hip.so (the C ABI of include/emdee_hip.h).

There is no CPU path: if the shared library is missing this module raises, and every call
checks the int32 status and raises EmDeeError with emdee_last_error().
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMDEE_HIP_LIB", os.path.join(_HERE, "libemdee_hip.so"))

OK = 0
F32, F64 = 4, 8
LITERAL, CUTOFF = 0, 1


class EmDeeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libemdee_hip status %d: %s" % (code, message))
        self.code = code


class LJModelC(C.Structure):          # emdee_lj_model
    _fields_ = [("rc2", C.c_double), ("rs2", C.c_double), ("inv_delta2", C.c_double)]


class LJAtomC(C.Structure):           # emdee_lj_atom == Julia LJAtom (src/lennard_jones.jl:15-18)
    _fields_ = [("half_sigma", C.c_float), ("twice_sqrt_eps", C.c_float)]


_p = C.c_void_p
_i32, _i64, _dbl = C.c_int32, C.c_int64, C.c_double
_pp = C.POINTER(C.c_void_p)
_d3 = C.POINTER(C.c_double)
_i3 = C.POINTER(C.c_int32)

# name -> argtypes; every entry returns int32 status unless listed in _RESTYPES
SIGNATURES = {
    "emdee_version": [],
    "emdee_device_count": [C.POINTER(_i32)],
    "emdee_ctx_create": [_i32, _p, _pp],
    "emdee_ctx_destroy": [_p],
    "emdee_sync": [_p],
    "emdee_device_info": [_p, C.c_char_p, C.c_size_t, C.POINTER(_i32), C.POINTER(_i64)],
    "emdee_malloc": [_p, C.c_size_t, _pp],
    "emdee_free": [_p, _p],
    "emdee_memcpy_h2d": [_p, _p, _p, C.c_size_t],
    "emdee_memcpy_d2h": [_p, _p, _p, C.c_size_t],
    "emdee_memcpy_d2d": [_p, _p, _p, C.c_size_t],
    "emdee_memset": [_p, _p, _i32, C.c_size_t],
    "emdee_interaction": [_p, _i32, _p, LJModelC, LJAtomC, LJAtomC, _i32, _p, _p, _i32],
    "emdee_cells_create": [_p, _i32, _dbl, _dbl, _i32, _i32, _pp],
    "emdee_cells_update": [_p, _p],
    "emdee_cells_destroy": [_p],
    "emdee_cells_M": [_p, C.POINTER(_i32)],
    "emdee_cells_arrays": [_p, _pp, _pp, _pp, _pp],
    "emdee_nbr_create": [_p, _i32, _dbl, _i32, _pp],
    "emdee_nbr_destroy": [_p],
    "emdee_nbr_stats": [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32)],
    "emdee_nbr_count_pairs": [_p, C.POINTER(_i64)],
    "emdee_nbr_set_exclusions": [_p, _p, _i32],
    "emdee_nbr_set_pairs14": [_p, _p, _i32, _dbl],
    "emdee_md_set_exclusions": [_p, _p, _i32],
    "emdee_md_set_pairs14": [_p, _p, _i32, _dbl],
    "emdee_nbr_list": [_p, _p, _p, _i32],
    "emdee_md_nbr_list": [_p, _p, _p, _i32],
    "emdee_compute_nonbonded": [_p, _p, _p, _p, _p, _dbl, _p, LJModelC, _p, _i32, _i32],
    "emdee_compute_nonbonded_tiles": [_p, _p, _p, _p, _p, _dbl, _i32, LJModelC, _p, _i32, _i32, _i32],
    "emdee_compute_nonbonded_naive": [_p, _p, _p, _p, _p, _dbl, _i32, LJModelC, _p, _i32, _i32],
    "emdee_md_create": [_p, _d3, _d3, _i3, LJModelC, _dbl, _i32, _pp],
    "emdee_md_destroy": [_p],
    "emdee_md_set_state": [_p, _i32, _i32, _p, _p, _p, _p],
    "emdee_md_get_state": [_p, _p, _p, _p, _p, _p],
    "emdee_md_step": [_p, _i32, _dbl, _i32],
    "emdee_md_kick_drift": [_p, _dbl, _dbl],
    "emdee_md_forces": [_p, _i32, _i32],
    "emdee_md_kick": [_p, _dbl],
    "emdee_md_fused_step": [_p, _dbl, _dbl, _i32, C.POINTER(_i32)],
    "emdee_md_needs_rebuild": [_p, C.POINTER(_i32)],
    "emdee_md_rebuild": [_p],
    "emdee_md_pack_positions": [_p, _p, _p, _i32, _d3, _i32, _p],
    "emdee_md_unpack_ghosts": [_p, _p, _i32, _i32],
    "emdee_md_energies": [_p, _d3],
    "emdee_md_nbr_stats": [_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32)],
    "emdee_md_count_pairs": [_p, C.POINTER(_i64)],
    "emdee_md_profile": [_p, _i32],
    "emdee_md_kernel_time": [_p, _i32, C.POINTER(_dbl), C.POINTER(_i64)],
    "emdee_md_set_langevin": [_p, _dbl, _dbl, C.c_uint64, C.c_uint64],
    "emdee_md_set_langevin_ids": [_p, _p],
    "emdee_md_langevin_normals": [_p, C.c_uint64, C.c_uint64, _p, _i32, _p],
    "emdee_dd_unique_id": [_p],
    "emdee_dd_rccl_selftest": [_p, _i32],
    "emdee_dd_describe": [_d3, _i3, _dbl, _i32, C.POINTER(_i32), _p, _p, _p, C.POINTER(_i32), _p, _d3, _d3, _i3],
    "emdee_dd_create": [_p, _d3, _i3, _i32, _i32, _p, LJModelC, _dbl, _i32, _pp],
    "emdee_dd_destroy": [_p],
    "emdee_dd_set_atoms": [_p, _i32, _i32, _p, _p, _p, _p],
    "emdee_dd_load": [_p],
    "emdee_dd_step": [_p, _i32, _dbl, _i32],
    "emdee_dd_energies": [_p, _d3],
    "emdee_dd_counts": [_p, _i32, C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32)],
    "emdee_dd_get_state": [_p, _i32, _p, _p, _p, _p],
    "emdee_dd_engine": [_p, _i32, _pp],
    "emdee_dd_set_langevin": [_p, _dbl, _dbl, C.c_uint64, C.c_uint64],
    "emdee_dd_stats": [_p, C.POINTER(_i64)],
    "emdee_dd_rebuild_stats": [_p, C.POINTER(_i64)],
    "emdee_dd_phase_times": [_p, C.POINTER(_dbl)],
    "emdee_dd_set_overlap": [_p, _i32],
    "emdee_last_error": [],
}
_RESTYPES = {"emdee_last_error": C.c_char_p}

_lib = None


def load():
    """dlopen libemdee_hip.so and declare every prototype. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)              # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int32)
    _lib = lib
    return lib


def last_error():
    msg = load().emdee_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status):
    if status != OK:
        raise EmDeeError(status, last_error())


def call(name, *args):
    check(getattr(load(), name)(*args))


def model_c(model):
    return LJModelC(float(model.rc2), float(model.rs2), float(model.inv_delta2))
