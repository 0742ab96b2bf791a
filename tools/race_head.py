#!/usr/bin/env python3
"""Diagnostic: does head_kernel (through eae_op_head_ce) give the same result when other kernels are co-resident?"""
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
B, L, Cc = 512, 64, 10
g = torch.Generator(device="cpu").manual_seed(5)
z = torch.randn((B, L), generator=g).to(dev)
w1 = (torch.randn((128, L), generator=g) * 0.2).to(dev); b1 = (torch.randn(128, generator=g) * 0.1).to(dev)
w2 = (torch.randn((Cc, 128), generator=g) * 0.2).to(dev); b2 = (torch.randn(Cc, generator=g) * 0.1).to(dev)
lab = torch.randint(0, Cc, (B,), generator=g).to(dev)
nsc = lib.eae_op_head_scratch_floats(B, L, Cc)
r4 = lambda n: (n + 3) // 4 * 4
NG = r4(128 * L) + 128 + r4(128 * Cc) + r4(Cc)


def run_head(stream):
    scratch = torch.zeros(nsc, dtype=torch.float32, device=dev)
    logits = torch.empty((B, Cc), device=dev); dz = torch.empty((B, L), device=dev)
    grads = torch.zeros(NG, device=dev); loss2 = torch.zeros(2, device=dev)
    check(lib.eae_op_head_ce(C.c_void_p(stream.cuda_stream), G.ptr(z), G.ptr(w1), G.ptr(b1), G.ptr(w2), G.ptr(b2), G.ptr(lab), B, L, Cc,
                             G.ptr(logits), G.ptr(dz), G.ptr(grads), G.ptr(loss2), G.ptr(scratch), nsc))
    return logits, dz, grads, loss2, scratch


# background work: the decoder's igemm kernels (kind 1) or a torch matmul
mode = sys.argv[1] if len(sys.argv) > 1 else "igemm"
# igemm[:kind:src:epi:cin:cout:hin]
f = mode.split(":")
kind, smode, epi, ci, co, hin = (int(v) for v in (f[1:] + ["1", "1", "0", "128", "64", "8"][len(f) - 1:]))
mode = f[0]
ho = hin // 2 if kind == 0 else hin * 2
x = (torch.randn((B, hin, hin, ci), device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty((B, ho, ho, co), device=dev, dtype=torch.bfloat16)
w = (torch.randn((co, 9, ci), device=dev) * 0.1).to(torch.bfloat16); bias = torch.randn(co, device=dev)
nt = lib.eae_op_conv_s2_ntiles(kind, ci, B, hin, hin)
part = torch.zeros((2, co, nt), device=dev)
cf = torch.randn((4, ci), device=dev)
yprev = (torch.randn((B, ho, ho, co), device=dev)).to(torch.bfloat16); pcoef = torch.randn((4, co), device=dev)
A = torch.randn((2048, 2048), device=dev); Bm = torch.randn((2048, 2048), device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()


def background(n):
    for _ in range(n):
        if mode == "igemm":
            sr = G.src(0, x) if smode == 0 else G.src(1, x, None, cf)
            check(lib.eae_op_conv_s2(C.c_void_p(main.cuda_stream), kind, sr, ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out),
                                     G.ptr(part) if epi != 2 else None, epi, G.ptr(yprev) if epi == 1 else None, G.ptr(pcoef) if epi == 1 else None))
        elif mode == "matmul":
            torch.mm(A, Bm)


torch.cuda.synchronize()
ref = [t.cpu().numpy().copy() for t in run_head(main)]
torch.cuda.synchronize()
bad = 0
for rep in range(60):
    background(6)
    with torch.cuda.stream(side):
        res = run_head(side)
    background(6)
    torch.cuda.synchronize()
    got = [t.cpu().numpy() for t in res]
    names = ["logits", "dz", "grads", "loss2", "scratch"]
    diffs = [n for n, a, b in zip(names, ref, got) if not np.array_equal(a, b)]
    if diffs:
        bad += 1
        if bad <= 2:
            d = np.flatnonzero((ref[1] != got[1]).any(axis=1))
            print(f"rep {rep}: differs in {diffs}; dz rows {d[:8].tolist()}", flush=True)
print(f"mode {sys.argv[1] if len(sys.argv) > 1 else mode}: {bad}/60 concurrent runs differ from the serial reference")
