#!/usr/bin/env python3
"""Diagnostic: per-step s_memtime stamps of one weight-gradient workgroup (producer wave 4, consumer wave 0) at B=512, plus the
kernel's time with raw operands (no load transforms) and at other grid sizes; needs the -DEAE_STAMPS build
(tools/build_variant.sh stamps -DEAE_STAMPS; EAE_LIB_PATH=<pkg>/libeae_stamps.so)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
from eae_amd import _lib  # noqa: E402
from eae_amd._lib import check  # noqa: E402

lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
B = 512
dev = torch.device("cuda:0")
dbg = torch.zeros(512, dtype=torch.int64, device=dev)
has_stamps = hasattr(raw, "eae_debug_set_wgrad")


def bf(shape):
    return (torch.randn(shape, device=dev) * 0.5).to(torch.bfloat16)


def coef(n, c):
    t = torch.randn((n, c), device=dev) * 0.1
    t[0] = 1.0 + t[0]
    return t.contiguous()


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


scratch = torch.empty(6 * 1024 * 1024, device=dev)
# (cs, cb, Hs, conv-type?)  conv-type: small = dy (BNBWD), big = activation (BNRELU); deconv-type: small = activation, big = dOut
for (cs, cb, hs, convt) in ((64, 32, 16, True), (64, 32, 16, False), (128, 64, 8, True), (128, 64, 8, False), (256, 128, 4, True)):
    small = bf((B, hs, hs, cs)); small2 = bf((B, hs, hs, cs))
    big = bf((B, 2 * hs, 2 * hs, cb)); big2 = bf((B, 2 * hs, 2 * hs, cb))
    dw = torch.zeros((cs, cb, 3, 3), device=dev)
    if convt:
        ss = G.src(2, small, small2, coef(3, cs)); bs = G.src(1, big, None, coef(4, cb))
    else:
        ss = G.src(1, small, None, coef(4, cs)); bs = G.src(2, big, big2, coef(3, cb))
    call = lambda: check(lib.eae_op_wgrad_s2(G.stream(), ss, bs, cs, cb, B, hs, hs, G.ptr(scratch), scratch.numel(), G.ptr(dw)))
    t = timeit(call)
    line = f"wgrad {cs}x{cb} Hs{hs} {'conv' if convt else 'deconv'}-type: {t:6.1f} us (+reduce)"
    if cs == 64:
        rs_, rb_ = G.src(0, small), G.src(0, big)
        t0 = timeit(lambda: check(lib.eae_op_wgrad_s2(G.stream(), rs_, rb_, cs, cb, B, hs, hs, G.ptr(scratch), scratch.numel(), G.ptr(dw))))
        line += f" | RAW operands {t0:6.1f} us"
    print(line, flush=True)
    def stamps(tag, fn):
        dbg.zero_()
        raw.eae_debug_set_wgrad(C.c_void_p(dbg.data_ptr()), 5)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        raw.eae_debug_set_wgrad(C.c_void_p(0), 0)
        t_ = dbg.cpu().tolist()
        for k in (2, 3):
            p = t_[4 * k: 4 * k + 5]; c = t_[256 + 2 * k: 256 + 2 * k + 3]
            print(f"   {tag} step {k}: producer stage {p[1]-p[0]:5d} issue {p[2]-p[1]:5d} barrier {p[4]-p[2]:5d} = {p[4]-p[0]:5d} | consumer mfma {c[1]-c[0]:5d} barrier {c[2]-c[1]:5d}")
    if has_stamps and cs == 64:
        rs_, rb_ = G.src(0, small), G.src(0, big)
        rawcall = lambda: check(lib.eae_op_wgrad_s2(G.stream(), rs_, rb_, cs, cb, B, hs, hs, G.ptr(scratch), scratch.numel(), G.ptr(dw)))
        stamps("RAW/RAW", rawcall)
        for mask, nm in ((1, "no stage"), (2, "no re-issue"), (3, "no stage, no re-issue"), (4, "no mfma"), (7, "barriers only")):
            raw.eae_debug_set_wgrad_exp(mask)
            tt = timeit(call)
            print(f"   exp [{nm}]: {tt:6.1f} us", flush=True)
            stamps(nm, call)
        raw.eae_debug_set_wgrad_exp(0)
    if has_stamps:
        dbg.zero_()
        raw.eae_debug_set_wgrad(C.c_void_p(dbg.data_ptr()), 5)
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        raw.eae_debug_set_wgrad(C.c_void_p(0), 0)
        t_ = dbg.cpu().tolist()
        for k in range(0, 8):
            p = t_[4 * k: 4 * k + 5]; c = t_[256 + 2 * k: 256 + 2 * k + 3]
            if not p[0] or not p[4]:
                break
            print(f"   step {k}: producer stage {p[1]-p[0]:5d} issue {p[2]-p[1]:5d} barrier {p[4]-p[2]:5d} = {p[4]-p[0]:5d} | consumer mfma {c[1]-c[0]:5d} barrier {c[2]-c[1]:5d}")
