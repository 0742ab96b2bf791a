#!/usr/bin/env python3
"""Diagnostic: which gradient tensors differ between repeated identical gradient steps at B=512 (bitwise reproducibility)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as gu  # noqa: E402
import eae_amd  # noqa: E402
from eae_amd.engine import engine_for  # noqa: E402

B = int(os.environ.get("KB", 512))
x = torch.rand((B, 3, 64, 64), device="cuda", generator=torch.Generator("cuda").manual_seed(3))
y = torch.randint(0, 10, (B,), device="cuda")
names = None
ref = None
for rep in range(int(os.environ.get("REPS", 6))):
    torch.manual_seed(0)
    m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).to("cuda")
    e = engine_for(m, max_batch=B)
    e.grad_step(x, y, 35.0)
    torch.cuda.synchronize()
    g = e.grads.cpu().numpy().copy()
    import ctypes as C
    ws = {}
    for (kind, idx, nm, per) in ((5, 0, "dyu0", 8 * 8 * 128), (3, 0, "gu0", 8 * 8 * 128), (2, 0, "u0", 8 * 8 * 128), (5, 1, "dyu1", 16 * 16 * 64), (4, 3, "dyy3", 4 * 4 * 256)):
        buf = np.empty(B * per, np.uint16)
        n = e.lib.eae_debug_read(e.ctx, kind, idx, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        assert n == buf.nbytes, n
        ws[nm] = buf
    if rep == 0:
        ws0 = ws
    else:
        for nm in ws:
            d = ws[nm] != ws0[nm]
            if d.any():
                ch = 128 if nm in ("dyu0", "gu0", "u0") else 64 if nm == "dyu1" else 256
                idxs = np.nonzero(d)[0]
                print(f"   workspace {nm}: {int(d.sum())} of {d.size} differ; channels {np.unique(idxs % ch)[:24]} images {np.unique(idxs // (d.size // B))[:12]}")
    if names is None:
        names = [n for n, _ in m.named_parameters()]
        sizes = [p.numel() for _, p in m.named_parameters()]
    if ref is None:
        ref = g
        continue
    off = 0
    bad = []
    # the arena aligns every tensor to 16 bytes: walk it with the same rule
    for n, s in zip(names, sizes):
        a, b = ref[off:off + s], g[off:off + s]
        if not np.array_equal(a, b):
            bad.append((n, int((a != b).sum()), s, float(np.abs(a - b).max()), float(np.abs(a).max())))
            if a.size == 256 * 128 * 9:
                d = (a != b).reshape(256, 128, 9)
                print("   rows(cs) with diffs:", np.unique(np.nonzero(d)[0])[:40], "cols(cb):", np.unique(np.nonzero(d)[1])[:40], "taps:", np.unique(np.nonzero(d)[2]))
        off += (s + 3) // 4 * 4
    print(f"rep {rep}: {'identical' if not bad else bad}", flush=True)
