"""The C-ABI library loads and exports every symbol include/eae.h declares; host-only entry points behave (no GPU)."""
import ctypes as C
import os
import re

import pytest

import eae_amd
from eae_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from eae_amd import build
        build.build(verbose=False)
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "eae.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(eae_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 30
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), n
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert lib.eae_version() >= 100


def test_layout_matches_module_shells(lib):
    import torch
    for latent in (64, 128):
        cfg = _lib.EaeConfig(latent, 10, 64, 64, 8)
        poff = (C.c_longlong * 39)()
        boff = (C.c_longlong * 15)()
        assert lib.eae_ae_layout(C.byref(cfg), poff, boff) == 0
        m = eae_amd.SupervisedAutoencoder(latent)
        sizes = [p.numel() for p in m.parameters()]
        assert len(sizes) == 38
        for i, n in enumerate(sizes):
            assert poff[i] % 4 == 0 and poff[i + 1] - poff[i] >= n and poff[i + 1] - poff[i] < n + 4
        assert boff[14] == 2 * (32 + 64 + 128 + 256 + 128 + 64 + 32)
    poff = (C.c_longlong * 11)()
    boff = (C.c_longlong * 5)()
    assert lib.eae_mlp_layout(64, 10, poff, boff) == 0
    sizes = [p.numel() for p in eae_amd.MLP(64).parameters()]
    assert len(sizes) == 10 and all(poff[i + 1] - poff[i] >= n for i, n in enumerate(sizes))


def test_error_convention(lib):
    for latent in (0, 257):          # any width in 1..256 is accepted (padded to a multiple of 64 inside the engine)
        bad = _lib.EaeConfig(latent, 10, 64, 64, 8)
        rc = lib.eae_ae_layout(C.byref(bad), None, None)
        assert rc == -2 and b"latent_dim" in lib.eae_last_error()
    bad = _lib.EaeConfig(64, 10, 60, 64, 8)
    assert lib.eae_ae_layout(C.byref(bad), None, None) == -2
    with pytest.raises(_lib.EaeError):
        _lib.check(lib.eae_mlp_layout(0, 10, None, None))
