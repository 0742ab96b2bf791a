#!/usr/bin/env python3
"""Diagnostic: are library fp32 elementwise / reduction kernels (torch: add, addcmul, sum -- what an all-reduce does to a
gradient bucket) perturbed when the engine's MFMA kernels are co-resident on another stream?"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
from eae_amd import _lib  # noqa: E402
from eae_amd._lib import check  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B = 512
x3 = torch.rand((B, 3, 64, 64), device=dev); w3 = (torch.randn((32, 64), device=dev) * 0.1).to(torch.bfloat16)
out3 = torch.empty((B, 32, 32, 32), device=dev, dtype=torch.bfloat16); part3 = torch.zeros((2, 32, B * 8), device=dev); b3 = torch.randn(32, device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()


def background(n):
    for _ in range(n):
        check(lib.eae_op_edge_conv(C.c_void_p(main.cuda_stream), 0, G.ptr(x3), B, 64, 64, G.ptr(w3), G.ptr(b3), G.ptr(out3), G.ptr(part3), 0, None, None))


g = torch.Generator(device="cpu").manual_seed(3)
a = torch.randn(1316048, generator=g).to(dev); b = torch.randn(1316048, generator=g).to(dev)


def work():
    c = a + b
    d = torch.addcmul(c, a, b, value=0.5)
    e = d * 0.125 + c
    s = e.sum()
    f = torch.stack([a, b, c, d]).sum(0)
    return c, d, e, s.reshape(1), f


torch.cuda.synchronize()
ref = [t.cpu().numpy().copy() for t in work()]
bad = 0
for rep in range(60):
    background(6)
    with torch.cuda.stream(side):
        res = work()
    background(6)
    torch.cuda.synchronize()
    got = [t.cpu().numpy() for t in res]
    if any(not np.array_equal(r, q) for r, q in zip(ref, got)):
        bad += 1
print(f"torch elementwise/reduction kernels beside edge_conv: {bad}/60 runs differ from the serial reference")
