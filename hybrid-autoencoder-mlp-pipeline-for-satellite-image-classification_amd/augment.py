"""On-device input staging: the reference's loader-side transforms (R.md:211-234) as one HIP kernel.

`augment_batch(u8, train=True)` takes a uint8 HWC batch already on the device ([B,H,W,3], 12 KB/img over PCIe instead of
48 KB of fp32) and returns the fp32 NCHW tensor the encoder reads: RandomHorizontalFlip -> RandomCrop(H, padding=4) ->
ToTensor -> AddGaussianNoise(0, 0.03) in training mode, ToTensor in eval mode.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check
from .engine import _ptr, _stream, _require_gpu


def augment_batch(u8, train=True, noise_std=0.03, seed=0, step=0, params=None, noise=None):
    if u8.dtype != torch.uint8 or u8.dim() != 4 or u8.shape[3] != 3:
        raise RuntimeError(f"expected uint8 HWC batch [B,H,W,3], got {tuple(u8.shape)} {u8.dtype}")
    _require_gpu(u8.device)
    lib = _lib.load()
    u8 = u8.contiguous()
    b, h, w, _ = u8.shape
    out = torch.empty((b, 3, h, w), dtype=torch.float32, device=u8.device)
    if params is not None:
        params = params.to(device=u8.device, dtype=torch.int32).contiguous()
        if params.shape != (b, 3):
            raise RuntimeError("params must be int32 [B,3] = (flip, top, left)")
    if noise is not None:
        noise = noise.to(device=u8.device, dtype=torch.float32).contiguous()
        if noise.shape != out.shape:
            raise RuntimeError("noise must be [B,3,H,W]")
    check(lib.eae_augment(_stream(), _ptr(u8), _ptr(out), b, h, w, int(train), float(noise_std), int(seed) & (2**64 - 1),
                          int(step) & (2**64 - 1), _ptr(params), _ptr(noise)))
    return out
