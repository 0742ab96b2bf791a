"""CPU baseline port (TEST / BENCH INFRASTRUCTURE ONLY): the reference's module graph restated with torch CPU tensor ops.

Same arithmetic as the notebook's ``SupervisedAutoencoder`` (R.md:287-433) and its batch loop (R.md:646-654), written
against ``torch.nn.functional`` so that it runs multi-threaded on the host cores exactly like the reference's CPU path
does (the reference itself cannot travel to the GPU box).  Pinned to the golden vectors in
``tests/test_oracle_golden.py::test_torch_cpu_port``.  Used only by ``bench.py``'s ``cpu_baseline`` leg and by tests.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

ENC = ((0, 1, 3, 32), (3, 4, 32, 64), (6, 7, 64, 128), (9, 10, 128, 256))
DEC = ((1, 2, 256, 128), (4, 5, 128, 64), (7, 8, 64, 32), (10, None, 32, 3))


def _uniform(shape, bound, gen):
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def build(latent_dim=64, num_classes=10, seed=0, state=None):
    """Parameter / buffer dict with the reference's state-dict names.  Default init = torch's defaults
    (U(+-1/sqrt(fan_in)); ConvTranspose2d fan_in = Cout*9)."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def conv(name, shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        p[name + ".weight"] = _uniform(shape, b, g)
        p[name + ".bias"] = _uniform((shape[0] if "decoder." not in name or "decoder_input" in name else shape[1],), b, g)

    def bn(name, c):
        p[name + ".weight"] = torch.ones(c)
        p[name + ".bias"] = torch.zeros(c)
        p[name + ".running_mean"] = torch.zeros(c)
        p[name + ".running_var"] = torch.ones(c)
        p[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)

    for ci, bi, cin, cout in ENC:
        conv(f"enc.encoder.{ci}", (cout, cin, 3, 3), cin * 9)
        bn(f"enc.encoder.{bi}", cout)
    conv("enc.encoder.13", (latent_dim, 4096), 4096)
    conv("dec.decoder_input", (4096, latent_dim), latent_dim)
    for di, bi, cin, cout in DEC:
        conv(f"dec.decoder.{di}", (cin, cout, 3, 3), cout * 9)
        if bi is not None:
            bn(f"dec.decoder.{bi}", cout)
    conv("classifier.0", (128, latent_dim), latent_dim)
    conv("classifier.2", (num_classes, 128), 128)
    if state is not None:
        for k, v in state.items():
            p[k] = torch.as_tensor(v).clone()
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    return p


def _bn(p, name, x, train):
    if train:
        p[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"], p[name + ".weight"], p[name + ".bias"],
                        training=train, momentum=0.1, eps=1e-5)


def forward(p, x, train=True, head=True):
    a = x
    for ci, bi, _, _ in ENC:
        a = F.conv2d(a, p[f"enc.encoder.{ci}.weight"], p[f"enc.encoder.{ci}.bias"], stride=2, padding=1)
        a = F.relu(_bn(p, f"enc.encoder.{bi}", a, train))
    z = F.linear(a.flatten(1), p["enc.encoder.13.weight"], p["enc.encoder.13.bias"])
    a = F.linear(z, p["dec.decoder_input.weight"], p["dec.decoder_input.bias"]).unflatten(1, (256, x.shape[2] // 16, x.shape[3] // 16))
    for di, bi, _, _ in DEC:
        a = F.conv_transpose2d(a, p[f"dec.decoder.{di}.weight"], p[f"dec.decoder.{di}.bias"], stride=2, padding=1, output_padding=1)
        a = torch.sigmoid(a) if bi is None else F.relu(_bn(p, f"dec.decoder.{bi}", a, train))
    logits = None
    if head:
        logits = F.linear(F.relu(F.linear(z, p["classifier.0.weight"], p["classifier.0.bias"])), p["classifier.2.weight"],
                          p["classifier.2.bias"])
    return a, logits, z


def make_adam(p, lr, weight_decay=0.0):
    return torch.optim.Adam([v for v in p.values() if v.requires_grad], lr=lr, weight_decay=weight_decay)


def train_step(p, opt, x, y, alpha, head=True):
    """zero_grad, forward, alpha*MSE + CE, backward, Adam  (R.md:646-654). Returns the loss as a float."""
    opt.zero_grad()
    x_hat, logits, _ = forward(p, x, True, head)
    loss = alpha * F.mse_loss(x_hat, x)
    if head:
        loss = loss + F.cross_entropy(logits, y)
    loss.backward()
    opt.step()
    return float(loss.detach())
