#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of one workgroup of the wave-specialised igemm2 kernel (consumer wave 0, producer wave 4); needs the
-DEAE_STAMPS build (tools/build_variant.sh stamps -DEAE_STAMPS) and EAE_IGEMM2=2 EAE_IG_SMALL=0."""
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
dbg = torch.zeros(32768, dtype=torch.int64, device=dev)
CASES = ((0, 64, 128, 16, 1, 0), (0, 128, 256, 8, 1, 0), (1, 256, 128, 4, 0, 0), (1, 128, 64, 8, 1, 0))
for (kind, ci, co, hin, smode, epi) in CASES:
    x = (torch.randn((B, hin, hin, ci), device=dev) * 0.5).to(torch.bfloat16)
    ho = hin // 2 if kind == 0 else hin * 2
    out = torch.empty((B, ho, ho, co), device=dev, dtype=torch.bfloat16)
    w = (torch.randn((co, 9, ci), device=dev) * 0.1).to(torch.bfloat16); bias = torch.randn(co, device=dev)
    nt = lib.eae_op_conv_s2_ntiles(kind, ci, B, hin, hin)
    part = torch.zeros((nt, 2, co), device=dev)
    cf = torch.randn((4, ci), device=dev)
    nc = ci // 32
    for blk in (0, nt - 1):
        dbg.zero_()
        raw.eae_debug_set(C.c_void_p(dbg.data_ptr()), blk)
        for _ in range(3):
            check(lib.eae_op_conv_s2(G.stream(), kind, G.src(smode, x, None, cf), ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out), G.ptr(part), epi, None, None))
        torch.cuda.synchronize()
        t = dbg.cpu().tolist()
        if t[1] == 0:
            print(f"kind{kind} {ci}->{co}: no igemm2 stamps (one-role kernel ran)")
            continue
        base = min(t[0], t[100])
        msg = [f"kind{kind} {ci}->{co} in{hin} blk {blk}: consumer total {t[2]-t[0]} | wait chunk0 {t[1]-t[0]}"]
        for c in range(nc):
            msg.append(f"c{c}: mfma {t[9+3*c]-t[8+3*c]} bar {t[10+3*c]-t[9+3*c]}")
        msg.append(f"epilogue {t[2]-t[10+3*(nc-1)]}")
        msg.append(f"|| producer: issue {t[101]-t[100]} fold+wait {t[102]-t[101]} stage0 {t[103]-t[102]} bar {t[104]-t[103]}")
        for c in range(nc - 1):
            msg.append(f"p{c+1}: stage {t[105+3*c]-t[104+3*c]} bar {t[106+3*c]-t[105+3*c]}")
        print(" | ".join(msg))
