#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of one workgroup of the edge kernels (deconv4+loss, conv1 forward, deconv4 backward-data) at B=512;
needs the -DEAE_STAMPS build (tools/build_variant.sh stamps -DEAE_STAMPS, EAE_LIB_PATH=.../libeae_stamps.so)."""
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
dbg = torch.zeros(64, dtype=torch.int64, device=dev)


def bf(shape):
    return (torch.randn(shape, device=dev) * 0.5).to(torch.bfloat16)


def coef(n, c):
    t = torch.randn((n, c), device=dev) * 0.1
    t[0] = 1.0 + t[0]
    return t.contiguous()


x = torch.rand((B, 3, 64, 64), device=dev)
a3 = bf((B, 32, 32, 32)); wj = bf((16, 128)); b3 = torch.randn(3, device=dev)
g4 = torch.zeros((B, 64, 64, 4), device=dev, dtype=torch.bfloat16); lp = torch.zeros((B * 8, 4), device=dev)
cf = coef(4, 32)
for blk in (0, 2047, 4095):
    dbg.zero_()
    raw.eae_debug_set_edge(C.c_void_p(dbg.data_ptr()), blk)
    for _ in range(3):
        check(lib.eae_op_deconv4_loss(G.stream(), G.src(1, a3, None, cf), B, 32, 32, G.ptr(wj), G.ptr(b3), G.ptr(x), 1e-3, None, G.ptr(g4), G.ptr(lp)))
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    print(f"deconv4+loss blk {blk}: total {t[7]-t[0]} | loads+coef {t[1]-t[0]} stage {t[2]-t[1]} bar {t[3]-t[2]} mfma {t[4]-t[3]} tile {t[5]-t[4]} elementwise {t[6]-t[5]} reduce+store {t[7]-t[6]}")
wp = bf((32, 64)); bias = torch.randn(32, device=dev)
out = bf((B, 32, 32, 32)); part = torch.zeros((B * 8, 2, 32), device=dev)
for blk in (0, 4095):
    dbg.zero_()
    raw.eae_debug_set_edge(C.c_void_p(dbg.data_ptr()), blk)
    for _ in range(3):
        check(lib.eae_op_edge_conv(G.stream(), 0, G.ptr(x), B, 64, 64, G.ptr(wp), G.ptr(bias), G.ptr(out), G.ptr(part), 0, None, None))
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    print(f"conv1 fwd blk {blk}: total {t[19]-t[16]} | patch {t[17]-t[16]} mfma+tile {t[18]-t[17]} epilogue {t[19]-t[18]}")
yprev = bf((B, 32, 32, 32)); pc = coef(4, 32)
for blk in (0, 4095):
    dbg.zero_()
    raw.eae_debug_set_edge(C.c_void_p(dbg.data_ptr()), blk)
    for _ in range(3):
        check(lib.eae_op_edge_conv(G.stream(), 1, G.ptr(g4), B, 64, 64, G.ptr(wp), None, G.ptr(out), G.ptr(part), 1, G.ptr(yprev), G.ptr(pc)))
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    print(f"deconv4 bwd-data blk {blk}: total {t[19]-t[16]} | patch {t[17]-t[16]} mfma+tile {t[18]-t[17]} epilogue {t[19]-t[18]}")
