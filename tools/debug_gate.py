#!/usr/bin/env python3
"""Diagnostic: does a stalled caller stream make the side-stream gates time out (EAE_GATE_TIMEOUT_MS)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import eae_amd
from eae_amd.engine import engine_for

def stall(seconds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe = 2_000_000
    torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(probe); e1.record(); torch.cuda.synchronize()
    ms = max(e0.elapsed_time(e1), 1e-3)
    print("probe", probe, "ticks ->", ms, "ms")
    probe = int(probe * 10.0 / ms)
    e0.record(); torch.cuda._sleep(probe); e1.record(); torch.cuda.synchronize()
    ms = max(e0.elapsed_time(e1), 1e-3)
    print("probe", probe, "ticks ->", ms, "ms")
    n = int(seconds * 1e3 / ms) + 1
    t0 = time.perf_counter()
    for _ in range(n):
        torch.cuda._sleep(probe)
    print("enqueued", n, "sleeps in", time.perf_counter() - t0, "s (host)")

for ms in ("100", "1"):
    os.environ["EAE_GATE_TIMEOUT_MS"] = ms
    torch.manual_seed(0)
    m = eae_amd.SupervisedAutoencoder(64).cuda()
    eng = engine_for(m, max_batch=8)
    x = torch.rand(8, 3, 64, 64, device="cuda"); y = torch.randint(0, 10, (8,), device="cuda")
    eng.train_step(x, y, 35.0, 5e-3); torch.cuda.synchronize()
    print("bound", ms, "ms: clean step timeouts", eng.gate_timeouts())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    stall(0.8)
    eng.train_step(x, y, 35.0, 5e-3)
    e1.record()
    torch.cuda.synchronize()
    print("bound", ms, "ms: stalled step took", e0.elapsed_time(e1), "ms; timeouts", eng.gate_timeouts(), "loss", eng.loss_last.tolist())
