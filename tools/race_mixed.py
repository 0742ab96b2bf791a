#!/usr/bin/env python3
"""Diagnostic: (1) packed-FMA victim with padded register allocations beside the MFMA->VALU aggressor; (2) both in ONE kernel."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vic = C.CDLL(os.path.join(ROOT, "tools", "probe", "libvictim.so"))
vicv = C.CDLL(os.path.join(ROOT, "tools", "probe", os.environ.get("VICTIM_LIB", "libvictim.so")))     # victim kernels may come from another build
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr())
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
sink = torch.zeros(512 * 256 * 4, device=dev)
out = torch.zeros(64 * 256 * 4, device=dev)
IT = 3000
# reference of the plain pk victim, serial
vicv.victim_pk(C.c_void_p(main.cuda_stream), P(out), 64, IT, 0); torch.cuda.synchronize(); ref = out.cpu().numpy().copy()
for agg in (8,):
    bad = 0
    for rep in range(12):
        out.zero_(); torch.cuda.synchronize()
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, agg)
        vicv.victim_pk(C.c_void_p(side.cuda_stream), P(out), 64, IT, 0)
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, agg)
        torch.cuda.synchronize()
        bad += not np.array_equal(ref, out.cpu().numpy())
    print(f"control: plain pk victim beside aggressor {agg}: {bad}/12 runs differ", flush=True)
for pad in ():
    bad = 0
    vic.victim_pk_pad(C.c_void_p(main.cuda_stream), P(out), 64, IT, pad); torch.cuda.synchronize()
    refp = out.cpu().numpy().copy()
    print("pad", pad, "serial equals plain victim:", bool(np.array_equal(refp, ref)))
    for rep in range(20):
        out.zero_(); torch.cuda.synchronize()
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, 8)
        vic.victim_pk_pad(C.c_void_p(side.cuda_stream), P(out), 64, IT, pad)
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, 8)
        torch.cuda.synchronize()
        bad += not np.array_equal(refp, out.cpu().numpy())
    print(f"padded victim ({pad} extra live VGPRs) beside the aggressor: {bad}/20 runs differ")
for n in ():
    bad = 0
    vic.victim_pk_alloc(C.c_void_p(main.cuda_stream), P(out), 64, IT, n); torch.cuda.synchronize()
    refp = out.cpu().numpy().copy()
    for rep in range(12):
        out.zero_(); torch.cuda.synchronize()
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, 8)
        vic.victim_pk_alloc(C.c_void_p(side.cuda_stream), P(out), 64, IT, n)
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, 8)
        torch.cuda.synchronize()
        bad += not np.array_equal(refp, out.cpu().numpy())
    print(f"victim allocation raised to v{n}: serial==plain {bool(np.array_equal(refp, ref))}; {bad}/12 runs differ beside the aggressor", flush=True)
for form, name in ((0, "op_sel:[0,1,0] (low lane reads the high source register)"), (1, "high element copied to a low register first (no op_sel)")):
    bad = 0
    vic.victim_pk_form(C.c_void_p(main.cuda_stream), P(out), 64, IT, form); torch.cuda.synchronize()
    refp = out.cpu().numpy().copy()
    for rep in range(20):
        out.zero_(); torch.cuda.synchronize()
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, 8)
        vic.victim_pk_form(C.c_void_p(side.cuda_stream), P(out), 64, IT, form)
        for _ in range(4): vic.aggressor(C.c_void_p(main.cuda_stream), P(sink), 512, 2000, 8)
        torch.cuda.synchronize()
        bad += not np.array_equal(refp, out.cpu().numpy())
    print(f"v_pk_fma_f32 form A/B: {name}: serial==plain {bool(np.array_equal(refp, ref))}; {bad}/20 runs differ beside the aggressor", flush=True)
# in-kernel mix: 1024 blocks, even = victim (512 victim blocks -> compare the first 64 with ref pattern per block)
outm = torch.zeros(512 * 256 * 4, device=dev)
bad = 0
for rep in range(20):
    outm.zero_(); torch.cuda.synchronize()
    vic.mixed(C.c_void_p(main.cuda_stream), P(outm), P(sink), 1024, IT)
    torch.cuda.synchronize()
    got = outm.cpu().numpy().reshape(512, 1024)
    bad += not all(np.array_equal(got[b], ref.reshape(64, 1024)[0]) for b in range(512))
print(f"victim and aggressor workgroups inside one kernel: {bad}/20 runs differ")
