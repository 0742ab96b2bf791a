#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime stamps of one workgroup of the igemm kernel (needs a -DEAE_STAMPS build:
   EAE_EXTRA_FLAGS=-DEAE_STAMPS python <pkg>/build.py --force)."""
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
dbg = torch.zeros(32768, dtype=torch.int64, device=dev)     # [0,64): phase stamps; beyond: the per-workgroup stamps of EAE_STAMP_WG
names = ["loads issued+coef", "transform+LDSwrite", "barrier", "MFMA(last chunk)", "barrier", "tile+rows(all ph)", "stats reduce"]
CASES = ((0, 32, 64, 32, 1, 0), (0, 64, 128, 16, 1, 0), (0, 128, 256, 8, 1, 0), (1, 256, 128, 4, 0, 0), (1, 128, 64, 8, 1, 0), (1, 64, 32, 16, 1, 0),
         (1, 64, 32, 16, 2, 1), (0, 32, 64, 32, 2, 1))       # (kind, cin, cout, hin, source mode, epilogue)
for (kind, ci, co, hin, smode, epi) in CASES:
    x = (torch.randn((B, hin, hin, ci), device=dev) * 0.5).to(torch.bfloat16)
    x2 = (torch.randn((B, hin, hin, ci), device=dev) * 0.5).to(torch.bfloat16) if smode == 2 else None
    ho = hin // 2 if kind == 0 else hin * 2
    out = torch.empty((B, ho, ho, co), device=dev, dtype=torch.bfloat16)
    yprev = (torch.randn((B, ho, ho, co), device=dev) * 0.5).to(torch.bfloat16) if epi == 1 else None
    pcoef = torch.randn((4, co), device=dev) if epi == 1 else None
    w = (torch.randn((co, 9, ci), device=dev) * 0.1).to(torch.bfloat16); bias = torch.randn(co, device=dev)
    nt = lib.eae_op_conv_s2_ntiles(kind, ci, B, hin, hin)
    part = torch.zeros((nt, 2, co), device=dev)
    cf = torch.randn((4, ci), device=dev)
    for blk in (0, nt - 1):
        raw.eae_debug_set(C.c_void_p(dbg.data_ptr()), blk)
        for _ in range(3):
            check(lib.eae_op_conv_s2(G.stream(), kind, G.src(smode, x, x2, cf), ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out), G.ptr(part), epi,
                                     G.ptr(yprev), G.ptr(pcoef)))
        torch.cuda.synchronize()
        t = dbg.cpu().tolist()
        nc = ci // 32
        msg = [f"kind{kind} {ci}->{co} in{hin} src{smode} epi{epi} blk {blk}: total {t[7]-t[0]}", f"start->chunk0 stage begin {t[8]-t[0]}"]
        for c in range(nc):
            b = 8 + c * 4
            nxt = t[8 + (c + 1) * 4] if c + 1 < nc else t[4]
            msg.append(f"c{c}: stage {t[b+1]-t[b]} bar {t[b+2]-t[b+1]} issue {t[b+3]-t[b+2]} mfma(+bar) {nxt-t[b+3]}")
        msg.append(f"epilogue: barrier {t[5]-t[4]} tile+rows {t[6]-t[5]} stats {t[7]-t[6]}")
        np_ = 2 if kind == 1 else 1
        prev = t[5]
        for ps in range(np_):
            b = 40 + ps * 4
            msg.append(f"pass{ps}: tilewrite {t[b]-prev} barrier {t[b+1]-t[b]} mfma-stats {t[b+2]-t[b+1]} rows {t[b+3]-t[b+2]}")
            prev = t[b + 3]
        print(" | ".join(msg))
