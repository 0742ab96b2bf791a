#!/usr/bin/env python3
"""Diagnostic: LDS / register / FMA victims (tools/probe/victim.hip) on a side stream while igemm kernels run on the main one."""
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

if os.environ.get("EAE_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["EAE_LIB"])       # diagnostic build variants
lib = _lib.load()
vic = C.CDLL(os.path.join(ROOT, "tools", "probe", "libvictim.so"))
dev = torch.device("cuda:0")
B = 512
kind, ci, co, hin = 1, 128, 64, 8
ho = hin * 2
NOSTAT = len(sys.argv) > 1 and sys.argv[1] == "nostat"
AGG = sys.argv[1] if len(sys.argv) > 1 else "igemm"
x = (torch.randn((B, hin, hin, ci), device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty((B, ho, ho, co), device=dev, dtype=torch.bfloat16)
w = (torch.randn((co, 9, ci), device=dev) * 0.1).to(torch.bfloat16); bias = torch.randn(co, device=dev)
nt = lib.eae_op_conv_s2_ntiles(kind, ci, B, hin, hin)
part = torch.zeros((2, co, nt), device=dev)
cf = torch.randn((4, ci), device=dev)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()


scratch = torch.empty(4 * 1024 * 1024, dtype=torch.float32, device=dev)
dw = torch.zeros((128, 64, 3, 3), dtype=torch.float32, device=dev)
big = (torch.randn((B, 16, 16, 64), device=dev)).to(torch.bfloat16); small = (torch.randn((B, 8, 8, 128), device=dev)).to(torch.bfloat16)
x3 = torch.rand((B, 3, 64, 64), device=dev); w3 = (torch.randn((32, 64), device=dev) * 0.1).to(torch.bfloat16)
out3 = torch.empty((B, 32, 32, 32), device=dev, dtype=torch.bfloat16); part3 = torch.zeros((2, 32, B * 8), device=dev); b3 = torch.randn(32, device=dev)


sink = torch.zeros(512 * 256 * 4, device=dev)


def background(n):
    for _ in range(n):
        if AGG.startswith("syn"):
            vic.aggressor(C.c_void_p(main.cuda_stream), G.ptr(sink), 512, 2000 if int(AGG[3:]) < 5 else 300, int(AGG[3:]))
            continue
        if AGG == "wgrad":
            check(lib.eae_op_wgrad_s2(C.c_void_p(main.cuda_stream), G.src(0, small), G.src(0, big), 128, 64, B, 8, 8, G.ptr(scratch), scratch.numel(), G.ptr(dw)))
            continue
        if AGG == "edge":
            check(lib.eae_op_edge_conv(C.c_void_p(main.cuda_stream), 0, G.ptr(x3), B, 64, 64, G.ptr(w3), G.ptr(b3), G.ptr(out3), G.ptr(part3), 0, None, None))
            continue
        if AGG == "edge_nostat":
            check(lib.eae_op_edge_conv(C.c_void_p(main.cuda_stream), 0, G.ptr(x3), B, 64, 64, G.ptr(w3), G.ptr(b3), G.ptr(out3), None, 0, None, None))
            continue
        if NOSTAT:
            check(lib.eae_op_conv_s2(C.c_void_p(main.cuda_stream), kind, G.src(1, x, None, cf), ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out), None, 0, None, None))
            continue
        check(lib.eae_op_conv_s2(C.c_void_p(main.cuda_stream), kind, G.src(1, x, None, cf), ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out), G.ptr(part), 0, None, None))


report = torch.zeros(512, dtype=torch.int32, device=dev)
NB = 1 << 20
idx = torch.arange(NB, dtype=torch.int64)
gbuf = (((idx * 2654435761) & 0xffffffff) ^ 0x5bd1e995).to(torch.int64)
gbuf = torch.where(gbuf >= 2 ** 31, gbuf - 2 ** 32, gbuf).to(torch.int32).to(dev)
L = 64
wv = torch.randn(128 * L, device=dev); zv = torch.randn(64 * 8 * L, device=dev)
fout = torch.zeros(64 * 8 * 128, device=dev)
vic.victim_fma(C.c_void_p(main.cuda_stream), G.ptr(wv), G.ptr(zv), G.ptr(fout), 64, L, None)
torch.cuda.synchronize()
fref = fout.cpu().numpy().copy()
pk_out = [torch.zeros(64 * 256 * 4, device=dev) for _ in range(2)]
pk_ref = []
for mode_ in range(2):
    vic.victim_pk(C.c_void_p(main.cuda_stream), G.ptr(pk_out[mode_]), 64, 3000, mode_)
    torch.cuda.synchronize()
    pk_ref.append(pk_out[mode_].cpu().numpy().copy())
nbad_pk = [0, 0]
nbad_fma = 0
for rep in range(40):
    background(6)
    vic.victim_lds(C.c_void_p(side.cuda_stream), G.ptr(report), 64, 50 * 1024, 40)
    vic.victim_reg(C.c_void_p(side.cuda_stream), G.ptr(report), 64, 40)
    vic.victim_gload(C.c_void_p(side.cuda_stream), G.ptr(gbuf), NB, 4, G.ptr(report), 64)
    background(6)
    vic.victim_ldsread(C.c_void_p(side.cuda_stream), 200, G.ptr(report), 64)
    fout.zero_()
    torch.cuda.synchronize()
    background(6)
    vic.victim_fma(C.c_void_p(side.cuda_stream), G.ptr(wv), G.ptr(zv), G.ptr(fout), 64, L, G.ptr(report))
    background(6)
    torch.cuda.synchronize()
    for mode_ in range(2):
        pk_out[mode_].zero_()
    torch.cuda.synchronize()
    background(6)
    for mode_ in range(2):
        vic.victim_pk(C.c_void_p(side.cuda_stream), G.ptr(pk_out[mode_]), 64, 3000, mode_)
    background(6)
    torch.cuda.synchronize()
    for mode_ in range(2):
        gotp = pk_out[mode_].cpu().numpy()
        if not np.array_equal(pk_ref[mode_], gotp):
            nbad_pk[mode_] += 1
            if nbad_pk[mode_] <= 2:
                d = np.flatnonzero(pk_ref[mode_] != gotp)
                print(f"pk victim mode {mode_}: {d.size} differ, threads%64 {sorted(set(((d // 4) % 64).tolist()))[:20]} comps {sorted(set((d % 4).tolist()))}", flush=True)
    if not np.array_equal(fref, fout.cpu().numpy()):
        nbad_fma += 1
        if nbad_fma <= 3:
            d = np.flatnonzero(fref != fout.cpu().numpy())
            got = fout.cpu().numpy()
            print("fma victim differs at", d[:4].tolist(), "of", d.size, "ref", fref[d[:4]].tolist(), "got", got[d[:4]].tolist(),
                  "rows", sorted(set((d // 128).tolist()))[:10], "j%64 range", int((d % 64).min()), int((d % 64).max()), flush=True)
r = report.cpu().numpy().astype(np.uint32)
print(AGG, f"gload victim: {r[2]} bad threads; ldsread victim: {r[3]} bad threads")
for s_ in range(min(int(r[2]), 6)):
    print("   gload thread %d first bad index %d got %08x (%d bad)" % tuple(int(v) for v in r[128 + 4 * s_: 132 + 4 * s_]))
for s_ in range(min(int(r[3]), 6)):
    print("   ldsread thread %d (%d bad iterations)" % tuple(int(v) for v in r[256 + 2 * s_: 258 + 2 * s_]))
print(AGG, f"fma victim LDS image vs memory: {r[8]} bad w1 words, {r[9]} bad zt words")
print(AGG, f"pk victim (register-only): packed {nbad_pk[0]}/40, scalar {nbad_pk[1]}/40 runs differ")
print(AGG, f"lds victim: {r[0]} corrupted workgroups; reg victim: {r[1]} corrupted registers; fma victim: {nbad_fma}/40 runs differ")
for s in range(min(int(r[0]), 10)):
    print("   block %d first bad word %d got %08x (%d bad words)" % tuple(int(v) for v in r[4 + 4 * s: 8 + 4 * s]))
