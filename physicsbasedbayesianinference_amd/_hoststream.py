"""NumPy's global legacy random stream, drawn faster and bit for bit (csrc/hoststream.c).

`standard_normal(shape)` and `uniform(size)` return exactly what `np.random.standard_normal` /
`np.random.uniform(size=...)` would return from the current global state and leave that state exactly
where NumPy would have left it (key, position, cached gaussian) -- so they interleave freely with
np.random calls and `np.random.seed(s)` keeps meaning what it means in the reference
(src/ensemble.py:72-74,88-91, src/HMC.py:168).  The MT19937 words are generated sequentially, the
polar Box-Muller transform runs in parallel (OpenMP) with the libm NumPy itself calls.  Small requests
and builds without libpbbi_host.so go to np.random directly: same numbers, NumPy's speed.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpbbi_host.so")
MIN_FAST = 1 << 14   # below this many values NumPy's own loop is as fast as the set-up here
_lib = None
_tried = False


class _State(C.Structure):
    _fields_ = [("key", C.c_uint32 * 624), ("pos", C.c_int), ("has_gauss", C.c_int), ("gauss", C.c_double)]


def _load():
    global _lib, _tried
    if not _tried:
        _tried = True
        if os.environ.get("PBBI_NO_HOSTSTREAM") != "1" and os.path.exists(LIB_PATH):
            lib = C.CDLL(LIB_PATH)
            for name in ("pbbi_host_standard_normal", "pbbi_host_random_sample"):
                fn = getattr(lib, name)
                fn.argtypes = [C.POINTER(_State), C.c_void_p, C.c_int64]
                fn.restype = C.c_int
            lib.pbbi_host_scaled_normal.argtypes = [C.POINTER(_State), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
            lib.pbbi_host_scaled_normal.restype = C.c_int
            # the box's CPU share for one GPU is 16 cores (the affinity mask may show the whole host)
            lib.pbbi_host_set_threads(C.c_int(max(1, min(16, len(os.sched_getaffinity(0))))))
            _lib = lib
    return _lib


def available():
    return _load() is not None


def _draw(fn_name, n, out=None, scale=None):
    lib = _load()
    kind, key, pos, has_gauss, gauss = np.random.get_state()
    if kind != "MT19937":
        return None
    st = _State()
    C.memmove(st.key, np.ascontiguousarray(key, dtype=np.uint32).ctypes.data, 624 * 4)
    st.pos, st.has_gauss, st.gauss = int(pos), int(has_gauss), float(gauss)
    if out is None:
        out = np.empty(n, dtype=np.float64)
    if scale is not None:
        rc = lib.pbbi_host_scaled_normal(C.byref(st), out.ctypes.data, n, scale.ctypes.data, scale.size)
    else:
        rc = getattr(lib, fn_name)(C.byref(st), out.ctypes.data, n)
    if rc != 0:
        return None   # out of memory / busy: the state is untouched, NumPy draws instead
    np.random.set_state(("MT19937", np.frombuffer(st.key, dtype=np.uint32).copy(), st.pos, st.has_gauss, st.gauss))
    return out


def _writable_f64(out):
    return (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous and
            out.flags.writeable)


def scaled_normal_into(out, scale_per_column):
    """out[...] = np.random.standard_normal(out.shape) * scale_per_column (broadcast over the last axis),
    written in place -- Ensemble.setMomentum's draw (src/ensemble.py:88-91) straight into an upload
    buffer.  `out`: C-contiguous float64."""
    n = out.size
    sc = np.ascontiguousarray(scale_per_column, dtype=np.float64)
    if sc.shape != (out.shape[-1],) or not _writable_f64(out):
        raise ValueError("out must be a writable C-contiguous float64 array and scale one value per column")
    if n >= MIN_FAST and _load() is not None and _draw(None, n, out=out.reshape(-1), scale=sc) is not None:
        return out
    np.multiply(np.random.standard_normal(out.shape), sc, out=out)
    return out


def uniform_into(out):
    """out[...] = np.random.uniform(size=out.shape), in place (C-contiguous float64)."""
    if not _writable_f64(out):
        raise ValueError("out must be a writable C-contiguous float64 array")
    n = out.size
    if n >= MIN_FAST and _load() is not None and _draw("pbbi_host_random_sample", n, out=out.reshape(-1)) is not None:
        return out
    out[...] = np.random.uniform(size=out.shape)
    return out


def standard_normal(shape):
    """np.random.standard_normal(shape) on the global legacy stream."""
    n = int(np.prod(shape))
    if n >= MIN_FAST and _load() is not None:
        out = _draw("pbbi_host_standard_normal", n)
        if out is not None:
            return out.reshape(shape)
    return np.random.standard_normal(shape)


def uniform(size):
    """np.random.uniform(size=size) (low 0, high 1) on the global legacy stream."""
    n = int(np.prod(size))
    if n >= MIN_FAST and _load() is not None:
        out = _draw("pbbi_host_random_sample", n)
        if out is not None:
            return out.reshape(size)
    return np.random.uniform(size=size)
