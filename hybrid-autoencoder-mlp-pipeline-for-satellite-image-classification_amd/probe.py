"""CE/MSE ratio probe of the reference (R.md:501-519): N freshly initialised
SupervisedAutoencoder(latent_dim=128) models, one batch each, forward in TRAIN mode under no_grad (so BatchNorm uses batch
statistics and updates its running statistics), ratio = CrossEntropy / MSE.

One engine context serves all trials: the parameters are re-drawn in place with the modules' own `reset_parameters()` (the
reference's default init), the forward and both losses run in the engine's kernels, and the N ratios are read back once.
"""
import torch

from .engine import engine_for
from .modules import SupervisedAutoencoder


def _reinit(model):
    for m in model.modules():
        if m is not model and hasattr(m, "reset_parameters"):
            m.reset_parameters()
        if hasattr(m, "reset_running_stats"):
            m.reset_running_stats()


def ce_mse_ratio_probe(batches, n_models=1000, latent_dim=128, num_classes=10, device="cuda", same_batch=True, on_model=None):
    """batches: iterable of (imgs, labels); the reference takes `next(iter(train_loader))` for every trial (R.md:509), i.e. a
    fresh first batch of a shuffled loader -- here the iterable is cycled (same_batch=False) or its first batch reused.
    on_model(i, model), if given, is called after trial i's parameters are in place (tests capture them for the oracle).
    Returns a list of n_models floats."""
    device = torch.device(device)
    it = iter(batches)
    imgs, labels = next(it)
    model = SupervisedAutoencoder(latent_dim=latent_dim, num_classes=num_classes).to(device)
    model.train()
    eng = engine_for(model, max_batch=imgs.shape[0])
    out = torch.zeros((n_models, 3), dtype=torch.float32, device=device)
    with torch.no_grad():
        for i in range(n_models):
            if i:
                _reinit(model)
                eng.params_changed()
                if not same_batch:
                    try:
                        imgs, labels = next(it)
                    except StopIteration:
                        it = iter(batches)
                        imgs, labels = next(it)
            if on_model is not None:
                on_model(i, model)
            x = imgs.to(device, non_blocking=True)
            y = labels.to(device, non_blocking=True)
            eng.forward(x, y, train=True, head=True, alpha=1.0, want=(), accum=False)      # loss_last = (loss, mse, ce)
            out[i].copy_(eng.loss_last[:3])
    r = out.cpu()
    return (r[:, 2] / r[:, 1]).tolist()
