"""Device plumbing: contexts, array transfer (stands in for CUDA.cu / CuArray / Array(dev)).

PyTorch is used only for HBM allocations and stream handles; all compute is in libemdee_hip.so.
Julia's 3xN column-major matrices are (N, 3) C-contiguous tensors here: the same bytes.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


class Context:
    """emdee_ctx on one device, enqueueing on torch's current stream for that device."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("emdee contexts live on a GPU (torch device type 'cuda' is HIP on ROCm)")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        _lib.call("emdee_ctx_create", index, C.c_void_p(stream), C.byref(h))
        self.handle = h
        self.stream = stream

    def sync(self):
        _lib.call("emdee_sync", self.handle)

    def info(self):
        arch = C.create_string_buffer(64)
        cus, hbm = C.c_int32(), C.c_int64()
        _lib.call("emdee_device_info", self.handle, arch, 64, C.byref(cus), C.byref(hbm))
        return dict(arch=arch.value.decode(), cu_count=cus.value, hbm_bytes=hbm.value)


_contexts = {}


def context_for(device=None):
    """One context per (device, current stream)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if gpu_available() else None
    if device is None:
        raise _lib.EmDeeError(-3, "no GPU visible: libemdee_hip has no CPU path")
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(torch.device("cuda", index)).cuda_stream)
    if key not in _contexts:
        _contexts[key] = Context(torch.device("cuda", index))
    return _contexts[key]


def gpu_available():
    n = C.c_int32(0)
    _lib.call("emdee_device_count", C.byref(n))
    return n.value > 0


def precision_of(t):
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.float64:
        return _lib.F64
    raise TypeError("arrays must be float32 or float64, got %s" % t.dtype)


def cu(x, device=None):
    """CUDA.cu(x): host array -> device tensor (test/runtests.jl:22,25).  Unlike CUDA.cu it keeps
    float64 as float64 (cast explicitly for the reference's Float32).  LJAtom arrays become (N, 2)
    float32 tensors with the same 8-byte records."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if isinstance(x, torch.Tensor):
        return x.to(device).contiguous()
    x = np.asarray(x)
    if x.dtype.names == ("half_sigma", "twice_sqrt_eps"):
        x = np.ascontiguousarray(x).view(np.float32).reshape(-1, 2)
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def to_host(t):
    """Array(dev): device tensor -> numpy."""
    return t.detach().cpu().numpy()


def check_array(t, name, rows, cols=None, dtype=None, device=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a GPU tensor" % name)
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    want = (rows,) if cols is None else (rows, cols)
    if tuple(t.shape) != want:
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), want))
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s has dtype %s, expected %s" % (name, t.dtype, dtype))
    if device is not None and t.device != device:
        raise ValueError("%s is on %s, expected %s" % (name, t.device, device))
    return t
