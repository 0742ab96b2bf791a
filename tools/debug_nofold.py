#!/usr/bin/env python3
"""Diagnostic: config-5 shape, loss / gradient norm of one gradient step, bf16 and fp8 (run with and without EAE_NO_FOLD_FWD=1)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eae_amd
from eae_amd.engine import AEEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator(device="cuda"); g.manual_seed(99)
x = torch.rand((B, 3, 256, 256), generator=g, device="cuda"); y = torch.randint(0, 10, (B,), generator=g, device="cuda")
for quant in ("bf16", "fp8"):
    torch.manual_seed(5)
    m = eae_amd.SupervisedAutoencoder(latent_dim=256, num_classes=10, image_size=256).cuda().train()
    e = AEEngine(m, max_batch=B, quant=quant)
    if quant == "fp8":
        e.fp8_calibrate(x, y, 35.0)
        print("scales", e.fp8_scales())
    e.grad_step(x, y, 35.0)
    torch.cuda.synchronize()
    print(quant, "loss", e.loss_last.cpu().numpy()[:3], "gnorm", float(e.grads.norm()), "nan", int(torch.isnan(e.grads).sum()))
    del e, m
