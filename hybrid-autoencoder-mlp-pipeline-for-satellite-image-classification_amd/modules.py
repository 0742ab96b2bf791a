"""Module shells with the reference notebook's signatures and state-dict layout.

``Encoder`` (R.md:287-313), ``Decoder`` (R.md:361-389), ``SupervisedAutoencoder`` (R.md:416-433) and
``MLP`` (R.md:2549-2566) keep the reference's constructor arguments, attribute names (``.encoder``,
``.decoder_input``, ``.decoder``, ``.enc``, ``.dec``, ``.classifier``, ``.net``), ``forward`` return
values and the 59-/16-tensor state-dict key layout, so checkpoints interchange both ways.

The layer objects inside the ``nn.Sequential`` containers are *parameter holders only*: they are
created with the same torch initialisers in the same order as the reference (so the same
``torch.manual_seed`` gives bit-identical initial weights) but their ``forward`` is disabled --
all arithmetic runs in the HIP engine (``engine.py`` -> ``csrc/`` through the C ABI of
``include/eae.h``).  There is no CPU or torch-eager fallback: calling ``forward`` without a HIP
device and the built extension raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _no_forward(self, *a, **k):
    raise RuntimeError(
        f"{type(self).__name__} is a parameter holder of the MI355X HIP engine; its arithmetic is not "
        "available as a stand-alone torch op. Call the owning Encoder/Decoder/SupervisedAutoencoder/MLP.")


class ConvS2Params(nn.Conv2d):
    """weight [Cout,Cin,3,3], bias [Cout] of a 3x3 stride-2 pad-1 convolution (R.md:292)."""
    forward = _no_forward


class DeconvS2Params(nn.ConvTranspose2d):
    """weight [Cin,Cout,3,3], bias [Cout] of a 3x3 stride-2 pad-1 output_padding-1 transposed conv (R.md:370)."""
    forward = _no_forward


class BatchNorm2dParams(nn.BatchNorm2d):
    forward = _no_forward


class BatchNorm1dParams(nn.BatchNorm1d):
    forward = _no_forward


class LinearParams(nn.Linear):
    forward = _no_forward


class _Marker(nn.Module):
    """Parameter-less placeholder keeping the reference's Sequential indices (ReLU, Flatten, ...)."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    forward = _no_forward


def _engine():
    from . import engine
    return engine


class Encoder(nn.Module):
    """R.md:287-313. forward(x [B,3,H,W] fp32) -> z [B,latent_dim]."""

    def __init__(self, latent_dim, image_size=64):
        super().__init__()
        self.latent_dim = int(latent_dim)
        self.image_size = int(image_size)
        fmap = self.image_size // 16
        self.encoder = nn.Sequential(
            ConvS2Params(3, 32, 3, stride=2, padding=1), BatchNorm2dParams(32), _Marker("ReLU"),
            ConvS2Params(32, 64, 3, stride=2, padding=1), BatchNorm2dParams(64), _Marker("ReLU"),
            ConvS2Params(64, 128, 3, stride=2, padding=1), BatchNorm2dParams(128), _Marker("ReLU"),
            ConvS2Params(128, 256, 3, stride=2, padding=1), BatchNorm2dParams(256), _Marker("ReLU"),
            _Marker("Flatten"),
            LinearParams(256 * fmap * fmap, self.latent_dim),
        )

    def forward(self, x):
        return _engine().encoder_forward(self, x)


class Decoder(nn.Module):
    """R.md:361-389. forward(z [B,latent_dim]) -> x_hat [B,3,H,W] in (0,1)."""

    def __init__(self, latent_dim, image_size=64):
        super().__init__()
        self.latent_dim = int(latent_dim)
        self.image_size = int(image_size)
        fmap = self.image_size // 16
        self.decoder_input = LinearParams(self.latent_dim, 256 * fmap * fmap)
        self.decoder = nn.Sequential(
            _Marker("Unflatten(1,(256,h,w))"),
            DeconvS2Params(256, 128, 3, stride=2, padding=1, output_padding=1), BatchNorm2dParams(128), _Marker("ReLU"),
            DeconvS2Params(128, 64, 3, stride=2, padding=1, output_padding=1), BatchNorm2dParams(64), _Marker("ReLU"),
            DeconvS2Params(64, 32, 3, stride=2, padding=1, output_padding=1), BatchNorm2dParams(32), _Marker("ReLU"),
            DeconvS2Params(32, 3, 3, stride=2, padding=1, output_padding=1),
            _Marker("Sigmoid"),
        )

    def forward(self, z):
        return _engine().decoder_forward(self, z)


class SupervisedAutoencoder(nn.Module):
    """R.md:416-433. forward(x) -> (x_hat, logits, z)."""

    def __init__(self, latent_dim, num_classes=10, image_size=64):
        super().__init__()
        self.latent_dim = int(latent_dim)
        self.num_classes = int(num_classes)
        self.enc = Encoder(latent_dim, image_size)
        self.dec = Decoder(latent_dim, image_size)
        self.classifier = nn.Sequential(
            LinearParams(self.latent_dim, 128),
            _Marker("ReLU"),
            LinearParams(128, self.num_classes),
        )
        _engine().register_children(self, (self.enc, self.dec))

    def forward(self, x):
        return _engine().autoencoder_forward(self, x)


class MLP(nn.Module):
    """R.md:2549-2566. forward(x [B,input_dim]) -> logits [B,num_classes]."""

    def __init__(self, input_dim, num_classes=10):
        super().__init__()
        self.input_dim = int(input_dim)
        self.num_classes = int(num_classes)
        self.net = nn.Sequential(
            LinearParams(self.input_dim, 128),
            BatchNorm1dParams(128),
            _Marker("ReLU"),
            _Marker("Dropout(0.3)"),
            LinearParams(128, 64),
            BatchNorm1dParams(64),
            _Marker("ReLU"),
            LinearParams(64, self.num_classes),
        )

    def forward(self, x):
        from . import mlp_engine
        return mlp_engine.mlp_forward(self, x)
