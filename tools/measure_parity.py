#!/usr/bin/env python3
"""Measured deviations behind the tolerances written in tests/test_gpu_ae.py (diagnostic; prints, asserts nothing).

    per-tensor gradient relmax / norm ratio at b=8 (vs the bf16-emulating oracle and vs the fp32 golden),
    digest deviations at b=2/32/48/56 (gradd/* of tests/golden/ae_fwd_bwd_b*.npz),
    final weights / BN buffers after the 5 golden Adam steps.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
import gpu_util as G  # noqa: E402
from helpers import ae_state_np, load_state_np  # noqa: E402
from oracle import ae_numpy as O  # noqa: E402
import eae_amd  # noqa: E402
from eae_amd.engine import engine_for  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def model():
    torch.manual_seed(gu.AE_SEED)
    m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10)
    load_state_np(m, ae_state_np())
    return m.to("cuda")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def main():
    g = np.load(os.path.join(GOLD, "ae_fwd_bwd_b8.npz"))
    p = ae_state_np()
    x, y, alpha = g["x"], g["labels"], float(g["alpha"])
    m = model()
    eng = engine_for(m, max_batch=64)
    eng.grad_step(cu(x), cu(y), alpha)
    torch.cuda.synchronize()
    eng.expose_grads()
    out = O.ae_forward(p, x, train=True, quant="bf16")
    gq = O.ae_backward(p, out, x, y, alpha, quant="bf16")
    print("== b=8: per tensor  relmax(q) normratio(q) cos(q) | relmax(32) normratio(32) cos(32)")
    for name, prm in m.named_parameters():
        got = prm.grad.cpu().numpy().astype(np.float64)
        r32 = g[f"grad/{name}"].astype(np.float64)
        if np.abs(r32).max() < 1e-6:
            continue
        rq = gq[name].astype(np.float64)
        nr = lambda a, b: float(np.linalg.norm(a) / max(1e-30, np.linalg.norm(b)))
        print(f"{name:28s} {G.relmax(got, rq):.3e} {nr(got, rq):.4f} {G.cosine(got, rq):.5f} | {G.relmax(got, r32):.3e} {nr(got, r32):.4f} {G.cosine(got, r32):.5f}"
              f" | oracle-vs-32 relmax {G.relmax(rq, r32):.3e}")
    for b in (2, 32, 48, 56):
        g = np.load(os.path.join(GOLD, f"ae_fwd_bwd_b{b}.npz"))
        x, y = gu.make_images(b, int(g["seed"]))
        m = model()
        eng = engine_for(m, max_batch=64)
        eng.grad_step(cu(x), cu(y), float(g["alpha"]))
        torch.cuda.synchronize()
        eng.expose_grads()
        print(f"== b={b}: l2 ratio, sample relmax, sample cos (vs fp32 golden digests)")
        for name, prm in m.named_parameters():
            got = prm.grad.cpu().numpy()
            dg, smp = g[f"gradd/{name}/digest"], g[f"gradd/{name}/sample"]
            if dg[1] < 1e-6:
                continue
            d, s = gu.tensor_digest(got)
            print(f"{name:28s} l2 {d[1] / dg[1]:.4f}  sample relmax {G.relmax(s, smp):.3e} cos {G.cosine(s, smp):.5f}")
    for tag, head in (("joint", True), ("recon", False)):
        g = np.load(os.path.join(GOLD, f"ae_adam5_{tag}_b8.npz"))
        m = model()
        eng = engine_for(m, max_batch=64)
        a = float(g["alpha"]) if head else 1.0
        for step in range(5):
            x, y = gu.make_images(8, 200 + step)
            eng.train_step(cu(x), cu(y), a, float(g["lr"]), head=head)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        p0 = ae_state_np()
        print(f"== adam5 {tag}: final weights vs golden digest (l2 ratio; sample max-abs diff; update size = |final-init|max)")
        for k in sd:
            dk = f"final/{k}/digest"
            if dk not in g.files:
                continue
            dg, smp = g[dk], g[f"final/{k}/sample"]
            d, s = gu.tensor_digest(sd[k])
            upd = float(np.abs(np.asarray(sd[k], np.float64) - p0[k]).max()) if k in p0 else float("nan")
            print(f"{k:40s} l2 {d[1] / max(dg[1], 1e-30):.5f} sample maxabs {np.abs(s - smp).max():.3e}  (|upd|max {upd:.3e}, |w|max {np.abs(smp).max():.3e})")


if __name__ == "__main__":
    main()
