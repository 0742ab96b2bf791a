#!/usr/bin/env python3
"""Diagnostic: start/end s_memtime stamps of EVERY workgroup of an igemm launch (-DEAE_STAMPS build) -> residency timeline."""
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
raw = C.CDLL(_lib.LIB_PATH)
B = 512
dev = torch.device("cuda:0")
dbg = torch.zeros(64 + 3 * 8192, dtype=torch.int64, device=dev)
for (kind, ci, co, hin) in ((0, 32, 64, 32), (1, 64, 32, 16), (0, 64, 128, 16), (0, 128, 256, 8)):
    x = (torch.randn((B, hin, hin, ci), device=dev) * 0.5).to(torch.bfloat16)
    ho = hin // 2 if kind == 0 else hin * 2
    out = torch.empty((B, ho, ho, co), device=dev, dtype=torch.bfloat16)
    w = (torch.randn((co, 9, ci), device=dev) * 0.1).to(torch.bfloat16); bias = torch.randn(co, device=dev)
    nt = lib.eae_op_conv_s2_ntiles(kind, ci, B, hin, hin)
    part = torch.zeros((nt, 2, co), device=dev)
    cf = torch.randn((4, ci), device=dev)
    raw.eae_debug_set(C.c_void_p(dbg.data_ptr()), 0)
    for _ in range(3):
        check(lib.eae_op_conv_s2(G.stream(), kind, G.src(1, x, None, cf), ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out), G.ptr(part), 0, None, None))
    torch.cuda.synchronize()
    t = dbg.cpu().numpy()
    n = min(nt, 8192)
    st = t[64:64 + 2 * n:2].astype(np.int64); en = t[65:65 + 2 * n:2].astype(np.int64)
    ids = t[64 + 2 * 8192:64 + 2 * 8192 + n]
    xcc = (ids >> 32) & 0xf
    t0 = st.min(); st -= t0; en -= t0      # s_memrealtime: 10 ns ticks, device-wide
    dur = en - st
    print(f"kind{kind} {ci}->{co} in{hin}: {n} WGs (x-blocks); span {en.max()} ticks; WG duration min/med/max {dur.min()}/{int(np.median(dur))}/{dur.max()}")
    print("   start-time percentiles (0,25,50,75,90,100):", [int(np.percentile(st, p)) for p in (0, 25, 50, 75, 90, 100)])
    print("   end-time percentiles:", [int(np.percentile(en, p)) for p in (0, 25, 50, 75, 90, 100)])
    print("   per-XCC first start / last end:", [(int(st[xcc == x].min()), int(en[xcc == x].max())) for x in range(8) if (xcc == x).any()])
    xcc = (ids >> 32) & 0xf
    cu = ((ids & 0xffffffff) >> 8) & 0xf
    se = ((ids & 0xffffffff) >> 13) & 0x7
    print("   WGs per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist(), " distinct (xcc,se,cu):", len(set(zip(xcc.tolist(), se.tolist(), cu.tolist()))))
    # residency over time
    edges = np.linspace(0, en.max(), 13)
    res = [(int(e), int(((st <= e) & (en > e)).sum())) for e in edges]
    print("   resident WGs at t:", res)
