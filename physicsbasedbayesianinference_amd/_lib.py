"""ctypes binding of libpbbi.so (C ABI: include/pbbi.h).

The product has no CPU path: if the shared library is missing or a call fails,
this module raises -- it never substitutes a host computation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PBBI_LIB: load another build of the same ABI (diagnostic builds, e.g. tools/stamp_probe.py)
LIB_PATH = os.environ.get("PBBI_LIB") or os.path.join(_HERE, "libpbbi.so")

OK = 0
F64, F32 = 0, 1
LEAPFROG, STORMER_VERLET = 0, 1
COMPAT_P_FROM_OLDQ = 1
KDK_FMA = 2
BETA_ACCEPT = 4
PER_CHAIN_STEPS = 8
UTURN_STOP = 16
DRAW_F64 = 32            # momenta drawn in double precision (include/pbbi.h)
STREAM_MOMENTUM, STREAM_POSITION, STREAM_UNIFORM, STREAM_STEPS = 0, 1, 2, 3
STREAM_SWAP = 4
STREAM_DRAW_F64 = 0x100  # OR-ed into pbbi_philox_normal's rng_stream: the draw DRAW_F64 selects


class PbbiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpbbi error {code}: {msg}")
        self.code = code


class DevInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64), ("compute_units", C.c_int),
                ("hbm_bytes", C.c_int64), ("lds_bytes_per_block", C.c_int), ("clock_khz", C.c_int)]


_vp, _i, _i64, _u64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
_dp = C.POINTER(C.c_double)
_pp = C.POINTER(C.c_void_p)

# name -> argtypes; every function returns int status except the two noted below.
PROTOTYPES = {
    "pbbi_version": [],
    "pbbi_last_error": [],
    "pbbi_device_count": [C.POINTER(C.c_int)],
    "pbbi_device_info": [_i, C.POINTER(DevInfo)],
    "pbbi_potential_create_harmonic": [_i, _dp, _i, _i, _pp],
    "pbbi_potential_create_gauss_diag": [_i, _dp, _dp, _d, _i, _i, _pp],
    "pbbi_potential_create_gauss_dense": [_i, _dp, _dp, _d, _i, _i, _pp],
    "pbbi_potential_create_rosenbrock": [_i, _d, _d, _d, _i, _i, _pp],
    "pbbi_potential_create_custom": [C.c_char_p, _i, _dp, _i, _i, _i, _pp],
    "pbbi_potential_destroy": [_vp],
    "pbbi_potential_dim": [_vp],
    "pbbi_potential_dtype": [_vp],
    "pbbi_potential_device": [_vp],
    "pbbi_potential_eval": [_vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "pbbi_integrate": [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _vp],
    "pbbi_leapfrog": [_vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _vp],
    "pbbi_stormer_verlet": [_vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _vp],
    "pbbi_energy": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "pbbi_weights_ratio": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp],
    "pbbi_hmc_iter": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _i, _vp],
    "pbbi_hmc_iter_kt": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _i, _d, _vp],
    "pbbi_hmc_iter_dyn": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _i, _d, _vp],
    "pbbi_hmc_run_dyn": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _i, _i, _u64, _u64,
                         _u64, _d, _vp],
    "pbbi_philox_steps": [_u64, _u64, _u64, _i64, _i, _i, _vp, _vp],
    "pbbi_describe_run": [_vp, _i, _i64, _i64, _i, _i, _i, C.c_char_p, _i],
    "pbbi_replica_exchange": [_vp, _vp, _i64, _i, _i64, _vp, _i, _u64, _u64, _u64, _vp, _vp],
    "pbbi_hmc_run_gist": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _i, _i, _u64, _u64, _u64, _d, _vp],
    "pbbi_hmc_run": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _d, _i, _i, _i, _u64, _u64,
                     _u64, _d, _vp],
    "pbbi_philox_normal": [_u64, _i, _u64, _u64, _i, _i64, _i64, _d, _vp, _i, _i, _vp, _vp],
    "pbbi_philox_uniform": [_u64, _u64, _u64, _i64, _i, _i, _vp, _vp],
    "pbbi_transpose_sdn_to_dns": [_vp, _vp, _i, _i, _i64, _i, _i, _vp],
    "pbbi_sample_moments": [_vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp],
    "pbbi_chain_moments": [_vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp],
    "pbbi_chain_autocov": [_vp, _vp, _i, _i, _i64, _i, _i, _i, _vp, _vp],
    "pbbi_sample_covariance": [_vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp],
    "pbbi_reduce_min": [_vp, _i64, _i, _i, _vp, _vp],
    "pbbi_canonical_weights": [_vp, _i64, _d, _vp, _i, _i, _vp, _vp, _vp],
    "pbbi_scale_inverse": [_vp, _i64, _vp, _i, _i, _vp],
}

_lib = None


def load():
    """Load libpbbi.so (built in-tree by `python -m physicsbasedbayesianinference_amd.build`
    or __graft_entry__.build()).  Raises if it is absent: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -m physicsbasedbayesianinference_amd.build); this package has no CPU fallback")
    # Bind to the SAME HIP runtime instance as PyTorch (our device-buffer plumbing): torch ships
    # its own libamdhip64.so.7; importing it first makes the loader resolve libpbbi.so's
    # dependency to that already-loaded copy instead of mapping a second runtime.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table drift apart
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.pbbi_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def last_error():
    return load().pbbi_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        raise PbbiError(rc, last_error())


def call(name, *args):
    check(getattr(load(), name)(*args))


def device_info(device=0):
    info = DevInfo()
    call("pbbi_device_info", int(device), C.byref(info))
    return dict(name=info.name.decode(), arch=info.arch.decode(), compute_units=info.compute_units,
                hbm_bytes=info.hbm_bytes, lds_bytes_per_block=info.lds_bytes_per_block,
                clock_khz=info.clock_khz)
