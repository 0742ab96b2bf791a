#!/usr/bin/env python3
"""Diagnostic: bitwise run-to-run reproducibility of the fused train step at the bench shape (B=512), many repetitions."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eae_amd  # noqa: E402
from eae_amd.engine import engine_for  # noqa: E402

B = int(os.environ.get("KB", 512)); REPS = int(os.environ.get("REPS", 40)); STEPS = int(os.environ.get("STEPS", 6))
g = torch.Generator(device="cpu").manual_seed(1234)
x = torch.rand((B, 3, 64, 64), generator=g).cuda(); y = torch.randint(0, 10, (B,), generator=g).cuda()
ref = None; bad = 0
for r in range(REPS):
    torch.manual_seed(0)
    m = eae_amd.SupervisedAutoencoder(64, 10).cuda(); m.train()
    eng = engine_for(m, max_batch=B)
    for s in range(STEPS):
        eng.train_step(x, y, 35.0, 5e-3)
    torch.cuda.synchronize()
    p = eng.params.cpu().numpy().copy()
    if ref is None:
        ref = p
    elif not np.array_equal(ref, p):
        bad += 1
        d = np.flatnonzero(ref != p)
        print(f"rep {r}: {d.size} differing floats, first at {d[:4].tolist()}", flush=True)
    del eng, m
print(f"{REPS} repetitions x {STEPS} steps at B={B}: {bad} differ from the first")
sys.exit(1 if bad else 0)
