#!/usr/bin/env python3
"""Throughput of the grouped B=64 step (eae_group_train_step) against K engines stepped one after the other.  Diagnostic tool (GPU box).

    python tools/group_bench.py [K ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eae_amd  # noqa: E402
from eae_amd.engine import AEEngine, engine_for  # noqa: E402

B = int(os.environ.get("GB_BATCH", "64"))
steps = int(os.environ.get("GB_STEPS", "200"))
ks = [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16]
x = torch.rand((B, 3, 64, 64), device="cuda")
y = torch.randint(0, 10, (B,), device="cuda")
pre = int(os.environ.get("GB_PRE", "0"))          # contexts (and their streams) created before the group's: the streams' hardware queues depend on it
keep_pre = []
for i in range(pre):
    m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
    keep_pre.append((m, engine_for(m, max_batch=B)))
if os.environ.get("GB_PRE_DROP", "1") == "1":
    keep_pre.clear()
    torch.cuda.empty_cache()
for k in ks:
    engs = []
    for i in range(k):
        torch.manual_seed(100 + i)
        m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
        if "EAE_SIDE_STREAMS" not in os.environ:
            m._eae_side_streams = 2          # what train.fit_autoencoder_group builds
        engs.append((m, engine_for(m, max_batch=B)))
    es = [e for _, e in engs]
    xs, ys, al, lr = [x] * k, [y] * k, [35.0] * k, [1e-3] * k
    for _ in range(20):
        AEEngine.group_train_step(es, xs, ys, al, lr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        AEEngine.group_train_step(es, xs, ys, al, lr)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    bad = sum(1 for e in es if e.gate_timeouts())
    print(f"K {k:2d}: {k * steps * B / el:10.0f} img/s   {1e3 * el / steps:.4f} ms per group step   host enqueue {1e3 * t_host / steps:.4f} ms   gate timeouts {bad}", flush=True)
    del engs, es
    torch.cuda.empty_cache()
