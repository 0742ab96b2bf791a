#!/usr/bin/env python3
"""Per-kernel timing at the bench shapes (B=512, 64x64) through the per-op C ABI. Diagnostic tool."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
from eae_amd import _lib  # noqa: E402
from eae_amd._lib import check  # noqa: E402

lib = _lib.load()
B = int(os.environ.get("KB", 512))
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bf(shape):
    return (torch.randn(shape, device=dev) * 0.5).to(torch.bfloat16)


def coef(n, c):
    t = torch.randn((n, c), device=dev) * 0.1
    t[0] = 1.0 + t[0]
    return t.contiguous()


rows = []
for (cin, cout, h) in ((32, 64, 32), (64, 128, 16), (128, 256, 8)):
    for kind in (0, 1):
        if kind == 0:
            ci, co, hin = cin, cout, h
            x = bf((B, hin, hin, ci)); out = bf((B, hin // 2, hin // 2, co))
        else:
            ci, co, hin = cout, cin, h // 2
            x = bf((B, hin, hin, ci)); out = bf((B, hin * 2, hin * 2, co))
        w = bf((co, 9, ci)); bias = torch.randn(co, device=dev)
        nt = lib.eae_op_conv_s2_ntiles(kind, ci, B, hin, hin)
        part = torch.zeros((nt, 2, co), device=dev)
        cf = coef(4, ci)
        s = G.src(0, x) if (kind == 1 and ci == 256) else G.src(1, x, None, cf)
        t = timeit(lambda: check(lib.eae_op_conv_s2(G.stream(), kind, s, ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out), G.ptr(part), 0, None, None)))
        m = out.numel() // co
        flops = 2.0 * m * co * ci * (9 if kind == 0 else 2.25)
        byts = (x.numel() + out.numel()) * 2
        rows.append((f"{'conv' if kind == 0 else 'deconv'} {ci}->{co} in{hin} FWD", t, flops / t / 1e6, byts / t / 1e3))
    # wgrad conv-type
    small = bf((B, h // 2, h // 2, cout)); big = bf((B, h, h, cin))
    g2 = bf((B, h // 2, h // 2, cout))
    scratch = torch.empty(6 * 1024 * 1024, device=dev); dw = torch.zeros((cout, cin, 3, 3), device=dev)
    ss = G.src(2, g2, small, coef(3, cout)); bs = G.src(1, big, None, coef(4, cin))
    t = timeit(lambda: check(lib.eae_op_wgrad_s2(G.stream(), ss, bs, cout, cin, B, h // 2, h // 2, G.ptr(scratch), scratch.numel(), G.ptr(dw))))
    flops = 2.0 * B * (h // 2) ** 2 * 9 * cin * cout
    rows.append((f"wgrad {cout}x{cin} Hs{h // 2} (+reduce) g,y | y", t, flops / t / 1e6, (small.numel() * 2 + big.numel()) * 2 / t / 1e3))
    # the train step's forms (round 4): the gradient operand is the stored dy tensor
    s4 = G.src(4, g2)
    t = timeit(lambda: check(lib.eae_op_wgrad_s2(G.stream(), s4, bs, cout, cin, B, h // 2, h // 2, G.ptr(scratch), scratch.numel(), G.ptr(dw))))
    rows.append((f"wgrad {cout}x{cin} Hs{h // 2} conv-type dy | y", t, flops / t / 1e6, (small.numel() + big.numel()) * 2 / t / 1e3))
    sa = G.src(0, small) if cout == 256 else G.src(1, small, None, coef(4, cout))
    b4 = G.src(4, big)
    t = timeit(lambda: check(lib.eae_op_wgrad_s2(G.stream(), sa, b4, cout, cin, B, h // 2, h // 2, G.ptr(scratch), scratch.numel(), G.ptr(dw))))
    rows.append((f"wgrad {cout}x{cin} Hs{h // 2} deconv-type y | dy", t, flops / t / 1e6, (small.numel() + big.numel()) * 2 / t / 1e3))

x = torch.rand((B, 3, 64, 64), device=dev); wp = bf((32, 64)); bias = torch.randn(32, device=dev)
out = bf((B, 32, 32, 32)); part = torch.zeros((B * 8, 2, 32), device=dev)
t = timeit(lambda: check(lib.eae_op_edge_conv(G.stream(), 0, G.ptr(x), B, 64, 64, G.ptr(wp), G.ptr(bias), G.ptr(out), G.ptr(part), 0, None, None)))
rows.append(("edge_conv (conv1 fwd)", t, 2.0 * B * 1024 * 27 * 32 / t / 1e6, (x.numel() * 4 + out.numel() * 2) / t / 1e3))
a3 = bf((B, 32, 32, 32)); wj = bf((16, 128)); b3 = torch.randn(3, device=dev); xh = torch.empty_like(x)
g4 = torch.zeros((B, 64, 64, 4), device=dev, dtype=torch.bfloat16); lp = torch.zeros((B * 8, 4), device=dev)
t = timeit(lambda: check(lib.eae_op_deconv4_loss(G.stream(), G.src(1, a3, None, coef(4, 32)), B, 32, 32, G.ptr(wj), G.ptr(b3), G.ptr(x), 1e-3, None, G.ptr(g4), G.ptr(lp))))
rows.append(("deconv4+loss", t, 2.0 * B * 4096 * 288 * 3 / 4 / t / 1e6, (a3.numel() * 2 + x.numel() * 4 + g4.numel() * 2) / t / 1e3))
print(f"{'kernel':40s} {'us':>8s} {'TFLOP/s':>9s} {'GB/s(alg)':>10s}")
for r in rows:
    print(f"{r[0]:40s} {r[1]:8.1f} {r[2]:9.1f} {r[3]:10.0f}")
