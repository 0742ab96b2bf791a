"""Helpers for the GPU parity tests (torch is used for device memory only)."""
import ctypes as C

import numpy as np
import torch

from eae_amd import _lib


def dev():
    return torch.device("cuda:0")


def to_nhwc_bf16(a):
    """numpy NCHW fp32 -> device NHWC bf16"""
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev()).permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)


def from_nhwc(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu().numpy()


def f32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def src(mode, p0, p1=None, coef=None):
    return _lib.EaeSrc(ptr(p0), ptr(p1), ptr(coef), mode)


def relmax(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def cosine(a, b):
    a = a.ravel().astype(np.float64)
    b = b.ravel().astype(np.float64)
    return float((a * b).sum() / max(1e-30, np.sqrt((a * a).sum() * (b * b).sum())))
