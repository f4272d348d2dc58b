"""Device-buffer plumbing: PyTorch-ROCm tensors are used ONLY to own HBM buffers and to
name HIP streams; every kernel launch goes through the C ABI with raw pointers."""
import os

import numpy as np

_torch = None


def torch():
    global _torch
    if _torch is None:
        import torch as _t  # deferred: first import takes a while on a fresh box
        _torch = _t
    return _torch


def default_device():
    """Index of the HIP device this process drives: PBBI_DEVICE, else LOCAL_RANK
    (one process per GPU under torchrun), else torch's current device."""
    for var in ("PBBI_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    t = torch()
    if not t.cuda.is_available():
        raise RuntimeError("no HIP device visible: the ensemble-HMC kernels need an AMD GPU "
                           "(there is no CPU fallback)")
    return t.cuda.current_device()


def torch_dtype(np_dtype):
    t = torch()
    return {np.dtype("float64"): t.float64, np.dtype("float32"): t.float32,
            np.dtype("uint8"): t.uint8, np.dtype("int32"): t.int32}[np.dtype(np_dtype)]


def dev(device):
    return torch().device("cuda", int(device))


def empty(shape, np_dtype, device):
    return torch().empty(tuple(int(x) for x in shape), dtype=torch_dtype(np_dtype), device=dev(device))


def as_device(host_array, device, np_dtype):
    """Copy a NumPy array into a fresh contiguous device tensor (H2D over PCIe)."""
    arr = np.ascontiguousarray(host_array, dtype=np_dtype)
    return torch().from_numpy(arr).to(dev(device), non_blocking=False)


def to_numpy(tensor):
    return tensor.detach().cpu().numpy()


def stream_ptr(device):
    return torch().cuda.current_stream(dev(device)).cuda_stream


def synchronize(device):
    torch().cuda.synchronize(dev(device))
