"""ctypes binding of libsigp.so (include/sigp.h).  No CPU fallback: if the HIP library is missing or no
MI355X is visible, importing is fine but creating an engine raises -- the product path never routes
through NumPy."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsigp.so")

OK, NOT_SPD, BAD_ARG, HIP_ERROR = 0, 1, 2, 3
KERNEL_IDS = {"netdiffusion": 0, "rbf": 1, "matern52": 2}
KCLASS = {"kbuild": 0, "diag": 1, "trsm": 2, "update_small": 3, "syrk128": 4, "epilogue": 5, "small": 6, "mlii": 7}
MAX_RIDE = 127

_dp = C.POINTER(C.c_double)
_i64 = C.c_int64
_ip64 = C.POINTER(C.c_int64)
_ip32 = C.POINTER(C.c_int32)
_h = C.c_void_p

# every symbol include/sigp.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "sigp_version": (C.c_int, []),
    "sigp_runtime_info": (C.c_int, [C.c_char_p, _i64]),
    "sigp_create": (C.c_int, [C.POINTER(_h), C.c_int, C.c_int]),
    "sigp_destroy": (C.c_int, [_h]),
    "sigp_last_error": (C.c_char_p, [_h]),
    "sigp_set_train": (C.c_int, [_h, _dp, _i64, _i64, _i64, _dp]),
    "sigp_set_test": (C.c_int, [_h, _dp, _i64, _i64]),
    "sigp_kernel_build": (C.c_int, [_h, C.c_int, C.c_double, C.c_double]),
    "sigp_kernel_build_from_sigma": (C.c_int, [_h, _dp, _i64, C.c_double]),
    "sigp_potrf": (C.c_int, [_h, C.POINTER(_i64)]),
    "sigp_fit": (C.c_int, [_h, _dp, _dp]),
    "sigp_predict_ride": (C.c_int, [_h, _dp, _dp]),
    "sigp_predict": (C.c_int, [_h, _dp, _i64, _i64, _dp, _dp]),
    "sigp_fit_predict": (C.c_int, [_h, C.c_int, C.c_double, C.c_double, _dp, _i64, _dp, _dp, _dp]),
    "sigp_fit_batch": (C.c_int, [_h, _i64, C.c_int, _dp, _i64, _dp, _i64, _dp, _i64, _i64, _i64, _i64, _dp, _dp,
                                 C.c_int, _dp, _dp, _dp]),
    "sigp_batch_upload": (C.c_int, [_h, _i64, _dp, _i64, _dp, _i64, _dp, _i64, _i64, _i64, _i64]),
    "sigp_batch_reserve": (C.c_int, [_h, _i64, C.c_int]),
    "sigp_batch_run": (C.c_int, [_h, _i64, _i64, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp]),
    "sigp_small_upload": (C.c_int, [_h, _i64, _ip64, _ip64, _ip64, _ip32, _dp, _ip64, _dp, _ip64, _dp, _ip64]),
    "sigp_small_run": (C.c_int, [_h, _i64, _ip64, _dp, _dp, _dp, _dp, _dp, _i64]),
    "sigp_small_set_dweights": (C.c_int, [_h, _dp, _i64]),
    "sigp_small_run_grad": (C.c_int, [_h, _i64, _ip64, _dp, _dp, _dp, _dp, _dp, _i64]),
    "sigp_get_stat": (C.c_int, [_h, C.c_char_p, _dp]),
    "sigp_area_sums": (C.c_int, [_h, _dp, _i64, _i64, _dp, _ip32, _i64, _dp]),
    "sigp_area_level": (C.c_int, [_dp, _i64, _ip32, _i64, _i64, C.c_int32, C.c_double, C.c_int, _ip32, _ip64, _ip32, _ip64, _ip32, _i64, _ip64]),
    "sigp_host_nanmean": (C.c_double, [_dp, _i64]),
    "sigp_detrend": (C.c_int, [_h, _dp, _i64, _i64, _i64, _ip64, _dp, _dp]),
    "sigp_corr_tau": (C.c_int, [_h, _dp, _i64, _i64, _i64, C.c_double, _dp, _i64, _dp, _dp]),
    "sigp_get_alpha": (C.c_int, [_h, _dp]),
    "sigp_get_matrix": (C.c_int, [_h, C.c_int, _dp, _i64]),
    "sigp_nlml_grad": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _i64, C.c_int, C.POINTER(C.c_double), _dp]),
    "sigp_nlml_grad_batch": (C.c_int, [_h, _i64, _i64, C.c_int, _dp, C.c_int, _dp, _dp]),
    "sigp_dist_unique_id": (C.c_int, [C.c_void_p]),
    "sigp_dist_init": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    "sigp_dist_init_transport": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p]),
    "sigp_dist_init_transport2": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p, _i64]),
    "sigp_dist_fit": (C.c_int, [_h, C.c_int, C.c_double, C.c_double, _dp, _i64, _i64, C.c_int, _dp, _dp, _dp]),
    "sigp_dist_predict": (C.c_int, [_h, _dp, _i64, _i64, _dp, _dp]),
    "sigp_dist_shutdown": (C.c_int, [_h]),
    "sigp_num_blocks": (_i64, [_h]),
    "sigp_profile": (C.c_int, [_h, C.c_int]),
    "sigp_profile_get": (C.c_int, [_h, C.c_int, _dp, C.POINTER(_i64), _dp, _dp]),
    "sigp_profile_reset": (C.c_int, [_h]),
    "sigp_synchronize": (C.c_int, [_h]),
    "sigp_set_option": (C.c_int, [_h, C.c_char_p, _i64]),
}

_lib = None
_dbg = None
DEBUG_LIB_PATH = os.path.join(_HERE, "libsigp_debug.so")


class SigpError(RuntimeError):
    pass


def load(debug=False):
    """dlopen libsigp.so and bind every declared symbol.  Fails loudly when the HIP build is missing.
    ``debug=True`` (tools/ only) loads libsigp_debug.so instead: the same library plus the micro-benchmark entry
    points of include/sigp_debug.h; the product package never asks for it."""
    global _lib, _dbg
    debug = debug or os.environ.get("SIGP_USE_DEBUG_LIB") == "1"      # tools/ only: measurement switches through the ordinary GPR class
    if debug and _dbg is not None:
        return _dbg
    if not debug and _lib is not None:
        return _lib
    path = DEBUG_LIB_PATH if debug else LIB_PATH
    if not os.path.exists(path):
        raise SigpError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C seaiceextentforecasting_amd/csrc%s` (there is no CPU fallback)" % (path, " debug" if debug else ""))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # must precede HIP runtime initialisation to take effect
    # No torch here.  libsigp.so binds /opt/rocm's libamdhip64; PyTorch-ROCm ships its own HIP runtime, and a process that wants
    # BOTH must import torch before the first load() (torch cannot initialise on top of an already initialised runtime, the
    # other order works): bench.py and the multi-rank workers do; runtime_info() says which runtime serves the process.
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if debug:
        _dbg = lib
    else:
        _lib = lib
    return lib


def runtime_info():
    """Which HIP runtime serves this process (for error reports)."""
    buf = C.create_string_buffer(512)
    try:
        load().sigp_runtime_info(buf, 512)
    except Exception as e:           # noqa: BLE001
        return "unknown (%s)" % e
    return buf.value.decode(errors="replace")


# the caller-supplied transport of the sharded fit (include/sigp.h: sigp_transport)
BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p)
SCATTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p)          # (ctx, buf, chunk_bytes, root, stream)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)                 # (ctx, buf, chunk_bytes, stream)


class Transport(C.Structure):
    """sigp_transport.  scatter / allgather are needed only for set_option("dist_panel_split", 1) and may stay NULL otherwise."""
    _fields_ = [("ctx", C.c_void_p), ("device_buffers", C.c_int), ("bcast", BCAST_FN), ("allreduce", ALLREDUCE_FN), ("scatter", SCATTER_FN),
                ("allgather", ALLGATHER_FN)]


def ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    """pointer to an int64 / int32 NumPy array"""
    return a.ctypes.data_as(_ip64 if a.dtype == np.int64 else _ip32)


def f64(a, ndim=None):
    """C-contiguous float64 copy/view (the reference's X is a transposed view: SURVEY App. C-8)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError("expected %d-D array, got shape %s" % (ndim, a.shape))
    return a
