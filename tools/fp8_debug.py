"""Diagnostic for the fp8 variant: engine vs fp8 oracle vs bf16 oracle at 256x256, B=2 (forward tensors and gradients)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402
import gpu_util as G  # noqa: E402
from helpers import load_state_np  # noqa: E402
from oracle import ae_numpy as O  # noqa: E402
import eae_amd  # noqa: E402
from eae_amd.engine import AEEngine  # noqa: E402

b = 2
torch.manual_seed(6)
m = eae_amd.SupervisedAutoencoder(latent_dim=256, num_classes=10, image_size=256)
p = gu.perturb_bn({k: v.detach().numpy().copy() for k, v in m.state_dict().items()})
load_state_np(m, p)
m = m.to("cuda")
rng = np.random.default_rng(12)
x = rng.random((b, 3, 256, 256)).astype(np.float32)
y = rng.integers(0, 10, b).astype(np.int64)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
eng = AEEngine(m, max_batch=b, quant="fp8")
for it in range(int(os.environ.get("ITERS", "7"))):
    eng.fp8_calibrate(xd, yd, 35.0, iters=1)
    print("iter", it, {k: [int(np.log2(s)) for s in v] for k, v in eng.fp8_scales().items()})
sc = eng.fp8_scales()
xh, lg, z = eng.forward(xd, labels=yd, train=True, alpha=35.0)
eng.grad_step(xd, yd, 35.0)
torch.cuda.synchronize()
eng.expose_grads()
o8 = O.ae_forward(p, x, train=True, quant="fp8", scales=sc)
ob = O.ae_forward(p, x, train=True, quant="bf16")
xg, zg = xh.cpu().numpy(), z.cpu().numpy()
print("x_hat: vs fp8 oracle mean %.5f max %.4f | vs bf16 oracle mean %.5f max %.4f | oracles apart mean %.5f" % (
    np.abs(xg - o8["x_hat"]).mean(), np.abs(xg - o8["x_hat"]).max(), np.abs(xg - ob["x_hat"]).mean(), np.abs(xg - ob["x_hat"]).max(),
    np.abs(o8["x_hat"] - ob["x_hat"]).mean()))
print("z: rel vs fp8 %.4f vs bf16 %.4f oracles apart %.4f" % (G.relmax(zg, o8["z"]), G.relmax(zg, ob["z"]), G.relmax(o8["z"], ob["z"])))
g8 = O.ae_backward(p, o8, x, y, 35.0, quant="fp8", scales=sc)
gb = O.ae_backward(p, ob, x, y, 35.0, quant="bf16")
for name, prm in m.named_parameters():
    gg = prm.grad.cpu().numpy()
    if np.abs(g8[name]).max() == 0:
        continue
    print(f"{name:28s} cos8 {G.cosine(gg, g8[name]):.4f} cosb {G.cosine(gg, gb[name]):.4f} o8-ob {G.cosine(g8[name], gb[name]):.4f} "
          f"ratio8 {np.linalg.norm(gg) / np.linalg.norm(g8[name]):.3f}")
