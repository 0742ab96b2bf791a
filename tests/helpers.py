"""Test helpers: build the package's module shells with the golden weights (CPU), numpy views, etc."""
import numpy as np
import torch

import eae_amd
import golden_util as gu


def ae_state_np(latent=64, perturb=True):
    """Reference-identical initial state (torch seed pinned by tests/golden/init_digest.npz) as numpy dict."""
    torch.manual_seed(gu.AE_SEED)
    m = eae_amd.SupervisedAutoencoder(latent_dim=latent, num_classes=10)
    sd = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    return gu.perturb_bn(sd) if perturb else sd


def mlp_state_np(perturb=True):
    torch.manual_seed(gu.MLP_SEED)
    c = eae_amd.MLP(input_dim=64, num_classes=10)
    sd = {k: v.detach().numpy().copy() for k, v in c.state_dict().items()}
    return gu.perturb_bn(sd, seed=9) if perturb else sd


def load_state_np(module, sd):
    module.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return module


def digest_close(arr, digest, sample, rtol=1e-4, atol=1e-6):
    d, s = gu.tensor_digest(arr)
    scale = max(1.0, float(digest[1]))
    # the plain sum cancels heavily, so it is only a coarse check
    assert abs(d[0] - digest[0]) <= 50 * rtol * scale + atol, (d, digest)
    assert abs(d[1] - digest[1]) <= rtol * scale + atol, (d, digest)
    np.testing.assert_allclose(s, sample, rtol=rtol * 10, atol=rtol * max(1e-3, float(np.abs(sample).max())))
