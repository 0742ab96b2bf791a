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


def ce_mse_ratio_probe(batches, n_models=1000, latent_dim=128, num_classes=10, device="cuda", same_batch=True, on_model=None, concurrent=1):
    """batches: iterable of (imgs, labels); the reference takes `next(iter(train_loader))` for every trial (R.md:509), i.e. a
    fresh first batch of a shuffled loader -- here the iterable is cycled (same_batch=False) or its first batch reused.
    on_model(i, model), if given, is called after trial i's parameters are in place (tests capture them for the oracle).
    concurrent=K > 1: K model + engine-context pairs on K streams / host threads (train.run_concurrent); trial i runs on pair i % K.
    The parameter draws stay ONE sequence in trial order (they come from torch's global generator, like the reference's constructor
    calls): a turnstile hands the generator from trial i to trial i + 1, only the forwards overlap.
    Returns a list of n_models floats."""
    import threading
    from .train import run_concurrent
    device = torch.device(device)
    k = max(1, min(int(concurrent), n_models))
    batch_list = [next(iter(batches))] if same_batch else None
    it = None if same_batch else iter(batches)
    out = torch.zeros((n_models, 3), dtype=torch.float32, device=device)
    turn = threading.Condition()
    state = {"next": 0}
    pairs = []
    for j in range(k):                 # trial j's parameters are the constructor's own draws (trials 0..k-1, in order)
        model = SupervisedAutoencoder(latent_dim=latent_dim, num_classes=num_classes).to(device)
        model.train()
        if k >= 3:
            model._eae_side_streams = -1       # one stream per context: four contexts then run side by side (4 hardware queues)
        pairs.append(model)

    def batch_for(i):
        nonlocal it
        if same_batch:
            return batch_list[0]
        try:
            return next(it)
        except StopIteration:
            it = iter(batches)
            return next(it)

    def worker(j):
        def job():
            model = pairs[j]
            eng = None
            with torch.no_grad():
                for i in range(j, n_models, k):
                    with turn:                      # the generator (and the batch iterator) pass through the trials in order
                        turn.wait_for(lambda: state["next"] == i)
                        if i >= k:
                            _reinit(model)
                        imgs, labels = batch_for(i)
                        if on_model is not None:
                            on_model(i, model)
                        state["next"] = i + 1
                        turn.notify_all()
                    if eng is None:
                        eng = engine_for(model, max_batch=imgs.shape[0])
                    eng.params_changed()
                    x = imgs.to(device, non_blocking=True)
                    y = labels.to(device, non_blocking=True)
                    eng.forward(x, y, train=True, head=True, alpha=1.0, want=(), accum=False)      # loss_last = (loss, mse, ce)
                    out[i].copy_(eng.loss_last[:3])
        return job
    run_concurrent([worker(j) for j in range(k)], k, device)
    r = out.cpu()
    return (r[:, 2] / r[:, 1]).tolist()
