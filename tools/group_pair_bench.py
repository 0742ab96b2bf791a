#!/usr/bin/env python3
"""Two (or G) groups of K/G configurations each, stepped concurrently from G host threads on their own streams: one group's forward
(latency-bound, most CUs idle) beside the other's backward.  Diagnostic tool (GPU box).

    python tools/group_pair_bench.py [G groups] [members per group]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eae_amd  # noqa: E402
from eae_amd.engine import AEEngine, engine_for  # noqa: E402
from eae_amd import train as T  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(os.environ.get("GB_STEPS", "200"))
x = torch.rand((64, 3, 64, 64), device="cuda")
y = torch.randint(0, 10, (64,), device="cuda")
groups = []
for g in range(G):
    es = []
    for i in range(M):
        torch.manual_seed(100 + g * M + i)
        m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
        es.append((m, engine_for(m, max_batch=64)))
    groups.append(es)


def job_of(es, n):
    engs = [e for _, e in es]
    a = ([x] * M, [y] * M, [35.0] * M, [1e-3] * M)

    def job():
        for _ in range(n):
            AEEngine.group_train_step(engs, *a)
    return job


T.run_concurrent([job_of(es, 20) for es in groups], G, static=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
T.run_concurrent([job_of(es, steps) for es in groups], G, static=True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
bad = sum(1 for es in groups for _, e in es if e.gate_timeouts())
print(f"{G} groups x {M}: {G * M * steps * 64 / el:10.0f} img/s   {1e3 * el / steps:.4f} ms per round   gate timeouts {bad}", flush=True)
