#!/usr/bin/env python3
"""Diagnostic: where does a NaN weight of enc.conv2 stop propagating?  (GPU)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import eae_amd
from eae_amd.engine import AEEngine
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
eng = AEEngine(m, max_batch=b)
x = torch.rand(b, 3, 64, 64, device="cuda"); y = torch.randint(0, 10, (b,), device="cuda")
with torch.no_grad():
    m.enc.encoder[3].weight[5, 0, 0, 0] = float("nan")
eng.params_changed()
xh, lg, z = eng.forward(x, labels=y, train=True, alpha=35.0)
torch.cuda.synchronize()
print("loss", eng.loss_last.cpu().numpy())
for i in (1, 4, 7, 10):
    bn = m.enc.encoder[i]
    print("enc bn", i, "nan rm", int(torch.isnan(bn.running_mean).sum()), "nan rv", int(torch.isnan(bn.running_var).sum()), "of", bn.running_var.numel())
for i in (2, 5, 8):
    bn = m.dec.decoder[i]
    print("dec bn", i, "nan rm", int(torch.isnan(bn.running_mean).sum()), "nan rv", int(torch.isnan(bn.running_var).sum()))
print("z nan", int(torch.isnan(z).sum()), "of", z.numel(), "logits nan", int(torch.isnan(lg).sum()), "x_hat nan", int(torch.isnan(xh).sum()), "of", xh.numel())
