"""ctypes binding of libmfx.so (include/mfx.h) -- the only way into the HIP kernels.

There is no CPU fallback: if the library is missing, or no MI355X is visible,
every compute entry point raises.  The library is looked for in-tree
(mfrec_amd/libmfx.so, built by __graft_entry__.build() / mfrec_amd/build.py).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFX_LIB: diagnostic builds only (e.g. libmfx_diag.so); the product is always libmfx.so
LIB_PATH = os.environ.get("MFX_LIB") or os.path.join(_HERE, "libmfx.so")

MFX_OK, MFX_E_ARG, MFX_E_RANGE, MFX_E_NOMEM, MFX_E_HIP, MFX_E_RCCL, MFX_E_SINGULAR, MFX_E_STATE = \
    0, -1, -2, -3, -4, -5, -6, -7
UNIQUE_ID_BYTES = 128

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)

# every symbol include/mfx.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mfx_version": (C.c_int, []),
    "mfx_last_error": (C.c_char_p, []),
    "mfx_device_count": (C.c_int, []),
    "mfx_init": (C.c_int, [C.c_int]),
    "mfx_sync": (C.c_int, []),
    "mfx_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "mfx_device_mem_info": (C.c_int, [_lp, _lp]),
    "mfx_timer_start": (C.c_int, []),
    "mfx_timer_stop": (C.c_int, [_dp]),
    "mfx_last_compute_ms": (C.c_double, []),
    "mfx_schedule_levels": (C.c_int, [_ip, C.c_int64, C.c_int, C.c_int, _ip, _lp]),
    "mfx_schedule_levels_gpu": (C.c_int, [_ip, C.c_int64, C.c_int, C.c_int, _ip, _lp]),
    "mfx_schedule_permutation": (C.c_int, [_ip, C.c_int64, C.c_int64, _lp, _lp]),
    "mfx_schedule_pack": (C.c_int, [_ip, _dp, _ip, C.c_int64, C.c_int64, C.c_void_p, _lp]),
    "mfx_schedule_balanced_order": (C.c_int, [_ip, C.c_int64, C.c_int, C.c_int, C.c_uint64, _lp, _lp]),
    "mfx_schedule_balanced_order_gpu": (C.c_int, [_ip, C.c_int64, C.c_int, C.c_int, _lp, _lp]),
    "mfx_kmf_train": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                C.c_double, _dp, _dp, _ip, _dp, _dp, _dp, C.c_int64, C.c_int,
                                C.c_int, C.c_int, C.c_int, _dp]),
    "mfx_funk": (C.c_int, [C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double,
                           C.c_double, _dp, _dp, _ip, _dp, _dp, _dp, C.c_int64, C.c_int, C.c_int,
                           C.c_int, C.c_int, C.c_int, _dp, C.c_int64, _lp]),
    "mfx_funk_dense": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                 C.c_double, C.c_double, _dp, _dp, _ip, _dp, C.c_int, _dp,
                                 C.c_int64, C.c_int, C.c_int]),
    "mfx_funk_subloop": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _dp, _dp,
                                   _ip, _dp, _dp, C.c_int64, C.c_int, C.c_int, _dp]),
    "mfx_funk_predictor_subloop": (C.c_int, [C.c_int, C.c_int, C.c_double, _dp, _dp, _ip, _dp, _dp,
                                             C.c_int64, C.c_int, C.c_int]),
    "mfx_learned_bias": (C.c_int, [C.c_int, C.c_double, C.c_int, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _dp,
                                   _dp, _ip, _dp, _dp, _dp, C.c_int64, C.c_int, C.c_int, C.c_int,
                                   _dp, C.c_int64, _lp]),
    "mfx_funk_implicit": (C.c_int, [C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                    _dp, _dp, _dp, _ip, _dp, _ip, C.c_int64, _ip, _dp, _dp, C.c_int64, C.c_int,
                                    C.c_int, C.c_int, _dp, C.c_int64, _lp]),
    "mfx_als_wrmf": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _ip, C.c_int64, _ip, _ip, C.c_int64,
                               _ip, C.c_int, C.c_int, C.c_int, C.c_double]),
    "mfx_als_wrmf_sharded": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _ip, C.c_int64, _ip, _ip,
                                       C.c_int64, _ip, C.c_int, C.c_int, C.c_int, C.c_double]),
    "mfx_predict_rmse": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _ip, _dp, _dp, _dp, C.c_double,
                                   C.c_int64, C.c_int, C.c_int, _dp, _dp]),
    "mfx_topn_stats": (C.c_int, [_lp]),
    "mfx_topn_plan_create": (C.c_void_p, [C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_int, C.c_int, _lp, _ip]),
    "mfx_topn_plan_query": (C.c_int, [C.c_void_p, _ip, C.c_int, C.c_int, C.c_int, _ip, _dp]),
    "mfx_topn_plan_destroy": (None, [C.c_void_p]),
    "mfx_topn_batch": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_int, _ip, C.c_int,
                                 C.c_int, C.c_int, _lp, _ip, _ip, _dp]),
    "mfx_sgd_plan_create": (C.c_void_p, [_ip, _dp, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "mfx_sgd_plan_create2": (C.c_void_p, [_ip, _dp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mfx_sgd_plan_create3": (C.c_void_p, [_ip, _dp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mfx_sgd_plan_res_info": (C.c_int, [C.c_void_p, _lp]),
    "mfx_sgd_plan_schedule_builder": (C.c_int, [C.c_void_p]),
    "mfx_sgd_plan_destroy": (None, [C.c_void_p]),
    "mfx_sgd_plan_info": (C.c_int, [C.c_void_p, _lp]),
    "mfx_sgd_plan_set_storage": (C.c_int, [C.c_void_p, C.c_int]),
    "mfx_sgd_plan_set_dot": (C.c_int, [C.c_void_p, C.c_int]),
    "mfx_sgd_plan_kernel_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "mfx_sgd_set_factors": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "mfx_sgd_get_factors": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "mfx_sgd_run_kmf": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_int, C.c_int, _dp]),
    "mfx_sgd_profile_kmf": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                      C.c_double, _dp, _lp]),
    "mfx_sgd_run_funk": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                                   C.c_int64, _lp]),
    "mfx_sgd_run_learned_bias": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_int, _dp, C.c_int64, _lp]),
    "mfx_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mfx_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "mfx_comm_destroy": (C.c_int, []),
    "mfx_comm_barrier": (C.c_int, []),
    "mfx_comm_allreduce_max": (C.c_int, [_dp]),
    "mfx_sgd_run_funk_sharded": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, _dp,
                                           C.c_int64, _lp]),
    "mfx_sgd_run_kmf_sharded": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_double, _dp, _dp]),
    "mfx_sgd_run_kmf_fake_world": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_double,
                                             C.c_double, C.c_double, C.c_double, C.c_double, _dp]),
    "mfx_dsgd_create": (C.c_void_p, [_ip, _dp, C.c_int64, C.c_int, C.c_int, C.c_int, _lp, _lp, C.c_int, C.c_int,
                                     C.c_int, C.c_int]),
    "mfx_dsgd_get_order": (C.c_int, [C.c_void_p, _lp]),
    "mfx_dsgd_destroy": (None, [C.c_void_p]),
    "mfx_dsgd_info": (C.c_int, [C.c_void_p, _lp]),
    "mfx_dsgd_set_factors": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "mfx_dsgd_run_kmf": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                   C.c_double, _dp]),
    "mfx_dsgd_run_kmf_averaged": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                            C.c_double, C.c_double, _dp]),
    "mfx_dsgd_get_factors": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "mfx_dsgd_run_funk": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                    C.c_int, C.c_int, C.c_int, _dp, C.c_int64, _lp]),
}

_lib = None


class MfxError(RuntimeError):
    pass


def lib():
    """Load libmfx.so (once).  Raises -- loudly -- if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfxError(
                "%s not found: build the HIP library first (python -c 'import __graft_entry__ as g; "
                "g.build()' or python -m mfrec_amd.build).  There is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error():
    return lib().mfx_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a negative status to the exception type the reference's caller would see."""
    if rc == MFX_OK:
        return
    msg = "libmfx error %d: %s" % (rc, last_error())
    if rc in (MFX_E_ARG, MFX_E_RANGE):
        raise ValueError(msg)
    if rc == MFX_E_NOMEM:
        raise MemoryError(msg)
    if rc == MFX_E_SINGULAR:
        raise np.linalg.LinAlgError(msg)
    raise MfxError(msg)


def f64(a, name, ndim, shape=None, allow_none=False):
    """Mimic Cython's `np.ndarray[np.float64_t, ndim=N, mode="c"]` buffer validation
    (e.g. gd_estimator.pyx:698-701): wrong dtype/ndim -> ValueError, non-array -> TypeError."""
    if a is None:
        if allow_none:
            return None
        raise TypeError("Argument '%s' must not be None" % name)
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch for '%s', expected 'float64_t' but got '%s'"
                         % (name, a.dtype))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions for '%s' (expected %d, got %d)"
                         % (name, ndim, a.ndim))
    if not a.flags.c_contiguous:
        raise ValueError("ndarray '%s' is not C-contiguous" % name)
    if not a.flags.writeable:
        raise ValueError("buffer source array '%s' is read-only" % name)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("'%s' has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a.ctypes.data_as(_dp)


def i32(a, name, ndim, shape=None):
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.dtype != np.int32:
        raise ValueError("Buffer dtype mismatch for '%s', expected 'int32_t' but got '%s'"
                         % (name, a.dtype))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions for '%s' (expected %d, got %d)"
                         % (name, ndim, a.ndim))
    if not a.flags.c_contiguous:
        raise ValueError("ndarray '%s' is not C-contiguous" % name)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("'%s' has shape %s, expected %s" % (name, a.shape, tuple(shape)))
    return a.ctypes.data_as(_ip)


def ratings_args(ratings_index, ratings):
    """Validate the (ratings_index int32[nnz,2], ratings f64[nnz]) pair of every SGD kernel."""
    pi = i32(ratings_index, "ratings_index", 2)
    if ratings_index.shape[1] != 2:
        raise ValueError("ratings_index must have shape [nnz, 2]")
    pr = f64(ratings, "ratings", 1)
    if ratings.shape[0] != ratings_index.shape[0]:
        raise ValueError("ratings_index and ratings disagree on nnz")
    return pi, pr, int(ratings.shape[0])


def factor_args(u, v, dim):
    pu = f64(u, "u", 2)
    pv = f64(v, "v", 2)
    if u.shape[0] != dim or v.shape[0] != dim:
        raise ValueError("u/v must be feature-major [dim=%d, n]; got %s and %s"
                         % (dim, u.shape, v.shape))
    return pu, pv, int(v.shape[1]), int(u.shape[1])


def device_count():
    return lib().mfx_device_count()


def init(device=0):
    check(lib().mfx_init(int(device)))


def sync():
    check(lib().mfx_sync())


def device_name():
    buf = C.create_string_buffer(256)
    check(lib().mfx_device_name(buf, 256))
    return buf.value.decode()
