"""Exported ECS columns as PyTorch-ROCm tensors (zero copy).

Mirrors ``madrona::py::Tensor`` / ``to_torch`` of the reference
(include/madrona/py/utils.hpp:73-141, src/python/bindings.cpp:52-68): a
non-owning view of executor-owned device memory.  PyTorch-ROCm consumes the
pointer through ``__cuda_array_interface__`` (HIP device pointers are valid
there); the simulator object must outlive the tensor.
"""
from __future__ import annotations

import numpy as np

_TYPESTR = {
    np.dtype(np.uint8): "|u1", np.dtype(np.int8): "|i1", np.dtype(np.int16): "<i2",
    np.dtype(np.int32): "<i4", np.dtype(np.int64): "<i8", np.dtype(np.float16): "<f2",
    np.dtype(np.float32): "<f4",
}


class DeviceColumn:
    """Minimal ``__cuda_array_interface__`` provider for a raw device pointer."""

    def __init__(self, ptr: int, dtype: np.dtype, shape, owner=None):
        self._owner = owner  # keep the simulator alive
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape),
            "typestr": _TYPESTR[np.dtype(dtype)],
            "data": (int(ptr), False),
            "version": 3,
            "strides": None,
        }


def to_torch(sim, name: str, device_index: int = 0):
    """Tensor aliasing exported column ``name`` of ``sim`` (HIP backend: a
    device tensor on ``cuda:device_index``; reference CPU backend: a CPU tensor
    sharing the staging buffer)."""
    import torch

    _, dtype, dims, on_device = sim.tensor_meta(name)
    ptr = sim.tensor_ptr(name)
    if on_device:
        col = DeviceColumn(ptr, dtype, dims, owner=sim)
        return torch.as_tensor(col, device=torch.device("cuda", device_index))

    import ctypes
    nbytes = int(np.prod(dims)) * np.dtype(dtype).itemsize
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype).reshape(dims)
    return torch.from_numpy(arr)
