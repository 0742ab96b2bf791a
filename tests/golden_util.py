"""Deterministic helpers shared by tools/make_golden.py (fixture generation, runs only in the
build container where /root/reference exists) and the tests (which never read /root/reference)."""
import numpy as np

AE_SEED = 0
MLP_SEED = 1


def make_images(b, seed, h=64, w=64):
    """Synthetic EuroSAT-shaped batch: fp32 NCHW in [0,1) + int64 labels (loader contract)."""
    rng = np.random.default_rng(seed)
    x = rng.random((b, 3, h, w), dtype=np.float32)
    y = rng.integers(0, 10, size=(b,), dtype=np.int64)
    return x, y


def make_latents(b, seed, d=64):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((b, d)).astype(np.float32)
    y = rng.integers(0, 10, size=(b,), dtype=np.int64)
    return x, y


def perturb_bn(sd, seed=7):
    """Give BatchNorm affine params / running stats non-trivial values (in place on a dict of numpy arrays).

    Default init (gamma=1, beta=0, rm=0, rv=1) would leave those code paths untested."""
    rng = np.random.default_rng(seed)
    for k in sorted(sd.keys()):
        if k.endswith("running_mean"):
            base = k[: -len("running_mean")]
            c = sd[k].shape[0]
            sd[base + "weight"] = (1.0 + 0.2 * rng.standard_normal(c)).astype(np.float32)
            sd[base + "bias"] = (0.1 * rng.standard_normal(c)).astype(np.float32)
            sd[base + "running_mean"] = (0.05 * rng.standard_normal(c)).astype(np.float32)
            sd[base + "running_var"] = (1.0 + 0.3 * rng.random(c)).astype(np.float32)
    return sd


def sample_idx(n, stride=97):
    return np.arange(0, n, stride)


def tensor_digest(a, stride=97):
    """(sum, l2, strided sample) digest of a big tensor for compact fixtures."""
    f = np.asarray(a, dtype=np.float64).ravel()
    return np.array([f.sum(), np.sqrt((f * f).sum())], np.float64), np.asarray(a, np.float32).ravel()[::stride].copy()


def is_prebn_bias(name):
    """Conv / deconv biases in front of a BatchNorm: analytically zero gradient (the engine writes exact zeros, the reference
    computes ~1e-9 rounding noise that Adam turns into a +-lr random walk: DESIGN.md section 5)."""
    parts = name.split(".")
    return name.endswith(".bias") and (
        name.startswith("enc.encoder.") and parts[2] in ("0", "3", "6", "9")
        or name.startswith("dec.decoder.") and parts[2] in ("1", "4", "7"))
