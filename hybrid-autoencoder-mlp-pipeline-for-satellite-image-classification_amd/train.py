"""fit / evaluate entry points restating the notebook's inline loops (there are no such functions in the reference;
SURVEY.md 8b):  AE grid + fit loop R.md:599-729, extract_features R.md:2498-2510, MLP grid + fit/test loop R.md:2611-2732,
final evaluate R.md:3171-3187.  Defaults = the notebook's literals.  Losses / accuracies are accumulated on the device
and read back once per epoch phase instead of the reference's per-step ``loss.item()`` (same sample-weighted means).

The loops drive a small "stepper" interface so the bookkeeping (weighted epoch means, early stopping, best tracking,
the reference's aliasing quirks, JSON schema) is testable without a GPU; the default steppers call the HIP engine.
"""
from __future__ import annotations

import json
import os

import torch

from .modules import SupervisedAutoencoder, MLP


# ------------------------------------------------------------------------------------------------ steppers
class AEStepper:
    """HIP-engine stepper for SupervisedAutoencoder: train_step / eval_step / read_loss."""

    def __init__(self, model, alpha, lr, head=True, max_batch=None, graph=None):
        from .engine import engine_for
        self.model, self.alpha, self.lr, self.head = model, float(alpha), float(lr), head
        self.eng = engine_for(model, max_batch=max_batch)
        self.eng.reset_optimizer()
        if graph is not None:
            self.eng.set_graph(graph)
        self.graph = bool(graph)
        self._stage = {}                     # graph replay: one fixed (imgs, labels) staging pair per batch size (a graph is keyed on its buffers)
        self.device = self.eng.device

    def begin(self):
        self.eng.reset_loss()

    def train_step(self, imgs, labels):
        if self.graph:
            b = int(imgs.shape[0])
            if b not in self._stage:
                self._stage[b] = (torch.empty_like(imgs), torch.empty_like(labels))
            xs, ys = self._stage[b]
            xs.copy_(imgs, non_blocking=True); ys.copy_(labels, non_blocking=True)
            imgs, labels = xs, ys
        self.eng.train_step(imgs, labels, self.alpha, self.lr, head=self.head)

    def eval_step(self, imgs, labels):
        self.eng.forward(imgs, labels=labels, train=False, head=self.head, alpha=self.alpha, want=(), accum=True)

    def end(self):
        loss, _, _, n, _ = self.eng.read_loss()
        self.eng.check_gates(sync=False)   # (read_loss has synchronised this stream) a timed-out side-stream gate must not go unnoticed
        return loss, n


class GroupAEStepper:
    """Stepper for SEVERAL SupervisedAutoencoder configurations trained in lockstep on the same batches (the grid of R.md:599-711):
    every step of the still-active members is ONE sequence of grouped launches (engine.AEEngine.group_train_step / group_eval_step,
    include/eae.h eae_group_train_step).  `active` = indices of the members that have not stopped early; the tile geometries stay
    those of the full group, so a member's arithmetic does not depend on which other members are still training."""

    def __init__(self, models, alphas, lrs, head=True, max_batch=None):
        from .engine import engine_for
        self.models, self.alphas, self.lrs, self.head = list(models), [float(a) for a in alphas], [float(v) for v in lrs], head
        self.engs = [engine_for(m, max_batch=max_batch) for m in self.models]
        for e in self.engs:
            e.reset_optimizer()
        self.mult = len(self.engs)
        self.device = self.engs[0].device

    def _pick(self, active):
        return [self.engs[k] for k in active], [self.alphas[k] for k in active], [self.lrs[k] for k in active]

    def begin(self, active):
        for k in active:
            self.engs[k].reset_loss()

    def train_step(self, imgs, labels, active):
        from .engine import AEEngine
        engs, alphas, lrs = self._pick(active)
        AEEngine.group_train_step(engs, [imgs] * len(engs), [labels] * len(engs), alphas, lrs, head=self.head, geometry_mult=self.mult)

    def eval_step(self, imgs, labels, active):
        from .engine import AEEngine
        engs, alphas, _ = self._pick(active)
        AEEngine.group_eval_step(engs, [imgs] * len(engs), [labels] * len(engs), alphas, head=self.head, geometry_mult=self.mult)

    def end(self, active):
        out = []
        for k in active:
            loss, _, _, n, _ = self.engs[k].read_loss()
            self.engs[k].check_gates(sync=False)
            out.append((loss, n))
        return out


class MLPStepper:
    def __init__(self, clf, lr, weight_decay=1e-4, max_batch=None):
        from .mlp_engine import mlp_engine_for
        self.clf, self.lr, self.wd = clf, float(lr), float(weight_decay)
        self.eng = mlp_engine_for(clf, max_batch=max_batch)
        self.eng.reset_optimizer()
        self.device = self.eng.device

    def begin(self):
        self.eng.reset_stats()

    def train_step(self, xb, yb):
        self.eng.train_step(xb, yb, self.lr, self.wd)

    def eval_step(self, xb, yb):
        self.eng.eval_step(xb, yb)

    def end(self):
        return self.eng.read_stats()      # (mean loss, accuracy, n)


def _to(t, device):
    return t if t.device == device else t.to(device, non_blocking=True)


def _first_batch_size(loader, default=64):
    bs = getattr(loader, "batch_size", None)
    return int(bs) if bs else default


# ------------------------------------------------------------------------------------------------ autoencoder
def fit_autoencoder(train_loader, val_loader, alpha, lr, latent_dim=64, num_classes=10, num_epochs=80, patience=15,
                    device="cuda", model=None, stepper=None, head=True, verbose=True, log=print, graph=None, side_streams=None):
    """One (alpha, lr) configuration of the reference's AE loop (R.md:619-697).

    Returns dict(model, train_curve, val_curve, best_val_loss, epochs).  As in the reference, `model` holds the weights of
    the LAST epoch run (R.md:705 keeps a live reference, not the best epoch's weights)."""
    if model is None and stepper is None:
        model = SupervisedAutoencoder(latent_dim=latent_dim, num_classes=num_classes).to(device)
        if side_streams is not None:
            model._eae_side_streams = side_streams      # read when the model's engine is first built (engine.engine_for)
    if stepper is None:
        stepper = AEStepper(model, alpha, lr, head=head, graph=graph,
                            max_batch=max(_first_batch_size(train_loader), _first_batch_size(val_loader)))
    dev = getattr(stepper, "device", None)
    counter, best_val_loss = 0, float("inf")
    train_curve, val_curve = [], []
    for epoch in range(num_epochs):
        if model is not None:
            model.train()
        stepper.begin()
        for imgs, labels in train_loader:
            if dev is not None:
                imgs, labels = _to(imgs, dev), _to(labels, dev)
            stepper.train_step(imgs, labels)
        train_loss, _ = stepper.end()
        train_curve.append(train_loss)
        if model is not None:
            model.eval()
        stepper.begin()
        with torch.no_grad():
            for imgs, labels in val_loader:
                if dev is not None:
                    imgs, labels = _to(imgs, dev), _to(labels, dev)
                stepper.eval_step(imgs, labels)
        val_loss, _ = stepper.end()
        val_curve.append(val_loss)
        if verbose:
            log(f"[AE α={alpha} LR={lr}] Epoch {epoch + 1} | TrainLoss={train_loss:.4f} | ValLoss={val_loss:.4f}")
        if val_loss < best_val_loss:          # strict improvement, R.md:690
            best_val_loss = val_loss
            counter = 0
        else:
            counter += 1
            if counter >= patience:
                if verbose:
                    log("Early stopping triggered.")
                break
    return {"model": model, "train_curve": train_curve, "val_curve": val_curve, "best_val_loss": best_val_loss,
            "epochs": len(train_curve)}


def fit_autoencoder_group(train_loader, val_loader, configs, latent_dim=64, num_classes=10, num_epochs=80, patience=15,
                          device="cuda", models=None, stepper=None, head=True, verbose=True, logs=None):
    """fit_autoencoder for SEVERAL (alpha, lr) configurations at once: the members share every batch of the two loaders (one pass of
    the loader per epoch for the whole group) and each keeps its own curves, best loss and early-stopping counter exactly as the
    reference's loop does for it alone (R.md:619-697); a member that stops early drops out, the others go on.

    Returns one fit_autoencoder-style dict per configuration.  logs: optional list of per-configuration line lists."""
    n = len(configs)
    if models is None and stepper is None:
        models = [SupervisedAutoencoder(latent_dim=latent_dim, num_classes=num_classes).to(device) for _ in range(n)]
        for m in models:
            m._eae_side_streams = 2              # a grouped step is fastest with two side streams (the engine's default is three)
    if stepper is None:
        stepper = GroupAEStepper(models, [a for a, _ in configs], [l for _, l in configs], head=head,
                                 max_batch=max(_first_batch_size(train_loader), _first_batch_size(val_loader)))
    dev = getattr(stepper, "device", None)
    lines = logs if logs is not None else [[] for _ in range(n)]
    counter, best = [0] * n, [float("inf")] * n
    train_curve, val_curve = [[] for _ in range(n)], [[] for _ in range(n)]
    active = list(range(n))
    for epoch in range(num_epochs):
        if not active:
            break
        if models is not None:
            for k in active:
                models[k].train()
        stepper.begin(active)
        for imgs, labels in train_loader:
            if dev is not None:
                imgs, labels = _to(imgs, dev), _to(labels, dev)
            stepper.train_step(imgs, labels, active)
        tr = stepper.end(active)
        if models is not None:
            for k in active:
                models[k].eval()
        stepper.begin(active)
        with torch.no_grad():
            for imgs, labels in val_loader:
                if dev is not None:
                    imgs, labels = _to(imgs, dev), _to(labels, dev)
                stepper.eval_step(imgs, labels, active)
        va = stepper.end(active)
        still = []
        for j, k in enumerate(active):
            alpha, lr = configs[k]
            train_curve[k].append(tr[j][0]); val_curve[k].append(va[j][0])
            if verbose:
                lines[k].append(f"[AE α={alpha} LR={lr}] Epoch {epoch + 1} | TrainLoss={tr[j][0]:.4f} | ValLoss={va[j][0]:.4f}")
            if va[j][0] < best[k]:            # strict improvement, R.md:690
                best[k] = va[j][0]; counter[k] = 0
                still.append(k)
            else:
                counter[k] += 1
                if counter[k] >= patience:
                    if verbose:
                        lines[k].append("Early stopping triggered.")
                else:
                    still.append(k)
        active = still
    return [{"model": None if models is None else models[k], "train_curve": train_curve[k], "val_curve": val_curve[k],
             "best_val_loss": best[k], "epochs": len(train_curve[k])} for k in range(n)]


_WORKER_STREAMS = {}       # (device, workers) -> the worker streams run_concurrent picked (reused: a context re-checks its side streams per caller's stream)


def run_concurrent(jobs, concurrent, device="cuda", static=False):
    """Run `jobs` (callables without arguments) on `concurrent` host threads, each with its OWN HIP stream current
    (torch.cuda.stream is thread-local), and return their results in job order.  This is how several small configurations share one
    GPU: at the reference's batch size 64 one train step is ~70 dependent launches of a few microseconds each, so a
    single stream leaves most of the 256 CUs idle and a single host thread cannot enqueue faster than the GPU drains; K engine contexts
    (own workspace, own side streams) stepped from K threads -- ctypes releases the GIL inside every libeae call -- overlap both.
    Worker k owns stream k and takes the next job when it is free; static=True: worker k runs jobs k, k + workers, ... (a job that is
    submitted again in a later call then finds the same stream current, and its contexts need not re-check their side streams)."""
    import threading
    dev = torch.device(device)
    if dev.type == "cuda" and not torch.cuda.is_available():
        dev = None
    nw = max(1, min(int(concurrent), len(jobs))) if jobs else 1
    results, errors = [None] * len(jobs), []
    nxt, nxt_lock = [0], threading.Lock()

    def take(k):
        if static:
            yield from range(k, len(jobs), nw)
            return
        while True:
            with nxt_lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(jobs):
                return
            yield i
    if dev is None or dev.type != "cuda":
        def cpu_worker(k):
            for i in take(k):
                try:
                    results[i] = jobs[i]()
                except BaseException as e:      # noqa: BLE001 -- re-raised below, in job order
                    errors.append((i, e))
        ths = [threading.Thread(target=cpu_worker, args=(k,)) for k in range(nw)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise min(errors, key=lambda t: t[0])[1]
        return results
    # The workers' streams: ROCm multiplexes a process's streams onto 4 hardware queues, and two streams on one queue run one after the
    # other.  Up to 4 workers get streams on pairwise DIFFERENT queues (candidates are drawn until they are: include/eae.h
    # eae_streams_share_queue); the streams are announced to the engine so that the contexts' side streams avoid them too, and they are
    # kept for the next call with the same number of workers (a context checks its side streams once per caller's stream).
    from . import _lib
    try:
        lib = _lib.load()
    except Exception:
        lib = None
    cache_key = (str(dev), int(concurrent))
    picked = _WORKER_STREAMS.get(cache_key)
    if picked is None:
        picked = []
        with torch.cuda.device(dev):
            for _ in range(max(1, int(concurrent))):
                cand = torch.cuda.Stream(device=dev)
                if lib is not None and int(concurrent) <= 4:
                    for _try in range(12):
                        if not any(lib.eae_streams_share_queue(cand.cuda_stream, p.cuda_stream) == 1 for p in picked):
                            break
                        cand = torch.cuda.Stream(device=dev)
                picked.append(cand)
        _WORKER_STREAMS[cache_key] = picked
    if lib is not None:
        for st in picked[:nw]:
            lib.eae_reserve_stream(st.cuda_stream, 1)

    def worker(k):
        st = picked[k]
        with torch.cuda.device(dev), torch.cuda.stream(st):
            for i in take(k):
                try:
                    results[i] = jobs[i]()
                    st.synchronize()
                except BaseException as e:      # noqa: BLE001 -- re-raised below, in job order
                    errors.append((i, e))
    try:
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(nw)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    finally:
        if lib is not None:
            for st in picked[:nw]:
                lib.eae_reserve_stream(st.cuda_stream, 0)
    if errors:
        raise min(errors, key=lambda t: t[0])[1]
    return results


def grid_search_autoencoder(train_loader, val_loader, alpha_values=(20, 25, 30, 35, 40),
                            lr_values=(1e-4, 2e-4, 5e-4, 1e-3, 2e-3, 5e-3, 1e-2, 5e-2, 1e-1), latent_dim=64, num_epochs=80,
                            patience=15, out_dir="models_best", device="cuda", verbose=True, log=print, fit_fn=None, concurrent=1,
                            grouped=0, group_fit_fn=None, concurrent_groups=1):
    """The reference's alpha x lr grid (R.md:599-729): trains every configuration, keeps the global best, writes
    `out_dir/AE_GLOBAL_BEST.pt` (plain state_dict) and `out_dir/validation_losses.json` (keys "alpha={a}, lr={lr}").

    concurrent=K > 1 trains K configurations at a time on the one GPU (run_concurrent): every configuration is an independent
    engine context on its own stream.  With the default `fit_fn` the models are built in grid order on the CALLING thread before the
    workers start, so under `torch.manual_seed(s)` every configuration starts from the parameters the sequential grid gives it and
    its curves and final weights are bitwise those of the same configuration trained alone -- provided the loaders' order does not
    depend on torch's global generator (a shuffling DataLoader draws its per-iterator seed from it, in whatever order the worker
    threads reach it: give such a loader its own `generator=`); a custom `fit_fn` that builds its own model has to seed it itself
    (tests/test_gpu_grid.py covers both).  The log lines of a configuration are emitted together, in grid order, and the global
    best is chosen in grid order with the reference's strict `<` -- the same winner as the sequential loop.

    grouped=K > 1 (takes precedence) trains the grid K configurations at a time IN LOCKSTEP (fit_autoencoder_group): the K members step
    on the same batches with ONE sequence of grouped launches per step -- at the reference's batch size 64 that is what fills the GPU
    (bench.py `configs.grid_b64`) -- and the loaders are walked once per epoch for the whole group.  Each
    member's arithmetic is bitwise that of the configuration trained alone on the same batches with the group's tile geometries
    (tests/test_gpu_grid.py); what differs from the sequential grid is the data order when the loader shuffles: the members of a
    group see the same permutation per epoch instead of consecutive draws from the generator.

    concurrent_groups=2 (with grouped=K) runs two groups at a time from two host threads (run_concurrent), every context with ONE side
    stream: 2 x 2 streams are the GPU's four hardware queues, and one group's forward pass -- a chain of latency-bound kernels that
    leaves most CUs idle -- runs beside the other group's backward pass (bench.py `configs.grid_b64`: 16 configurations at 0.89-0.92 M
    images/s against 0.83-0.85 M for one group of 8).  The models are built in grid order on the calling thread, a configuration's
    results do not depend on what runs beside it."""
    os.makedirs(out_dir, exist_ok=True)
    fit_fn = fit_fn or fit_autoencoder
    results, best = {}, {"loss": float("inf"), "info": None, "state": None, "train": None, "val": None}
    grid = [(alpha, lr) for alpha in alpha_values for lr in lr_values]
    fitted = None
    if grouped and int(grouped) > 1:
        gfit = group_fit_fn or fit_autoencoder_group
        pairs = max(1, int(concurrent_groups or 1))
        chunks = [grid[g0:g0 + int(grouped)] for g0 in range(0, len(grid), int(grouped))]
        prebuilt = None
        if pairs > 1 and gfit is fit_autoencoder_group:
            # parameter initialisation draws from torch's global generator: in grid order, on this thread (as for `concurrent`)
            prebuilt = []
            for cfgs in chunks:
                ms = [SupervisedAutoencoder(latent_dim=latent_dim, num_classes=10).to(device) for _ in cfgs]
                for m in ms:
                    m._eae_side_streams = 1          # two groups x (caller's stream + one side stream) = the four hardware queues
                prebuilt.append(ms)

        def group_job(gi):
            def job():
                cfgs = chunks[gi]
                lines = [[] for _ in cfgs]
                extra = {"models": prebuilt[gi]} if prebuilt is not None else {}
                rs = gfit(train_loader, val_loader, cfgs, latent_dim=latent_dim, num_epochs=num_epochs, patience=patience, device=device,
                          verbose=verbose, logs=lines, **extra)
                out = []
                for r, ln in zip(rs, lines):
                    r = dict(r)
                    r["state"] = None if r.get("model") is None else {k: v.detach().cpu().clone() for k, v in r["model"].state_dict().items()}
                    r["model"] = None           # release the engine (workspace, streams) of a finished configuration
                    out.append((r, ln))
                if prebuilt is not None:
                    prebuilt[gi] = None
                return out
            return job
        done = run_concurrent([group_job(gi) for gi in range(len(chunks))], pairs, device) if pairs > 1 else [group_job(gi)() for gi in range(len(chunks))]
        fitted = [item for chunk in done for item in chunk]
    elif concurrent and int(concurrent) > 1:
        # parameter initialisation draws from torch's global generator: K worker threads would draw in timing-dependent order
        prebuilt = {}
        if fit_fn is fit_autoencoder:
            for (alpha, lr) in grid:
                m = SupervisedAutoencoder(latent_dim=latent_dim, num_classes=10).to(device)
                if int(concurrent) >= 3:
                    m._eae_side_streams = -1
                prebuilt[(alpha, lr)] = m

        def job_of(alpha, lr):
            def job():
                lines = []
                # Measured on MI355X at B=64 (bench.py `configs.grid_b64`): a process reaches the GPU through 4 hardware
                # queues.  Contexts with the default three streams each share them -- K = 2 / 4 / 8 / 16 reach 1.65x the single-configuration
                # rate and stay there -- while ONE stream per context lets four configurations run side by side: 487 K vs 352 K images/s
                # at K = 4.  hipGraph replay of each step is slower than eager (0.7-0.8x), more hardware queues or several processes far slower.
                extra = {"model": prebuilt.pop((alpha, lr))} if fit_fn is fit_autoencoder else {}
                r = fit_fn(train_loader, val_loader, alpha, lr, latent_dim=latent_dim, num_epochs=num_epochs, patience=patience,
                           device=device, verbose=verbose, log=lines.append, **extra)
                r = dict(r)
                # only the state_dict of a finished configuration is needed below: release its engine (workspace, streams) now
                r["state"] = None if r.get("model") is None else {k: v.detach().cpu().clone() for k, v in r["model"].state_dict().items()}
                r["model"] = None
                return r, lines
            return job
        fitted = run_concurrent([job_of(a, l) for a, l in grid], int(concurrent), device)
    for gi, (alpha, lr) in enumerate(grid):
            if verbose:
                log("\n=====================================")
                log(f"Training AE for α={alpha}, LR={lr}")
                log("=====================================")
            if fitted is not None:
                r, lines = fitted[gi]
                if verbose:
                    for ln in lines:
                        log(ln)
            else:
                r = fit_fn(train_loader, val_loader, alpha, lr, latent_dim=latent_dim, num_epochs=num_epochs, patience=patience,
                           device=device, verbose=verbose, log=log)
            results[(alpha, lr)] = r["best_val_loss"]
            if r["best_val_loss"] < best["loss"]:
                state = r.get("state")
                if state is None and r.get("model") is not None:
                    state = {k: v.detach().cpu().clone() for k, v in r["model"].state_dict().items()}
                best.update(loss=r["best_val_loss"], info=(alpha, lr), train=r["train_curve"], val=r["val_curve"], state=state)
                if verbose:
                    log("\nNew best AE")
                    log(f"   α={alpha}, LR={lr}, ValLoss={r['best_val_loss']:.4f}")
    best_path = os.path.join(out_dir, "AE_GLOBAL_BEST.pt")
    if best["state"] is not None:
        torch.save(best["state"], best_path)
    with open(os.path.join(out_dir, "validation_losses.json"), "w") as f:
        json.dump({f"alpha={a}, lr={lr}": float(v) for (a, lr), v in results.items()}, f, indent=4)
    return {"best_alpha": best["info"][0], "best_lr": best["info"][1], "best_val_loss": best["loss"], "best_path": best_path,
            "results": results, "best_train_curve": best["train"], "best_val_curve": best["val"]}


def extract_features(loader, encoder):
    """Same name and signature as the reference (R.md:2498-2510): eval-mode encoder over a loader -> CPU (X [N,L], y [N])."""
    X_list, y_list = [], []
    encoder.eval()
    dev = next(encoder.parameters()).device
    with torch.no_grad():
        for imgs, labels in loader:
            imgs = imgs.to(dev, non_blocking=True)
            z = encoder(imgs)
            X_list.append(z.cpu())
            y_list.append(labels.cpu())
    return torch.cat(X_list, dim=0), torch.cat(y_list, dim=0)


# ------------------------------------------------------------------------------------------------ MLP
def fit_mlp(train_dl, val_dl, test_dl, lr, input_dim=64, num_classes=10, num_epochs=30, weight_decay=1e-4, device="cuda",
            clf=None, stepper=None, alias_best=True, verbose=True, log=print):
    """One learning rate of the reference's MLP loop (R.md:2619-2697): Adam(lr, weight_decay), CE, accuracy tracking,
    best-validation snapshot, test accuracy.

    alias_best=True reproduces the reference's `clf.state_dict().copy()` (R.md:2683): a shallow copy that aliases the
    live tensors, so the "best" state -- and the test accuracy measured after `load_state_dict` -- are those of the
    final epoch.  alias_best=False snapshots real copies of the best-validation epoch instead."""
    if clf is None and stepper is None:
        clf = MLP(input_dim=input_dim, num_classes=num_classes).to(device)
    if stepper is None:
        stepper = MLPStepper(clf, lr, weight_decay, max_batch=max(256, _first_batch_size(train_dl)))
    dev = getattr(stepper, "device", None)
    curves = {"train_acc": [], "val_acc": [], "train_loss": [], "val_loss": []}
    best_val_acc, best_state = 0, None

    def run(dl, train):
        stepper.begin()
        for xb, yb in dl:
            if dev is not None:
                xb, yb = _to(xb, dev), _to(yb, dev)
            (stepper.train_step if train else stepper.eval_step)(xb, yb)
        return stepper.end()

    for e in range(num_epochs):
        if clf is not None:
            clf.train()
        tr_loss, tr_acc, _ = run(train_dl, True)
        if clf is not None:
            clf.eval()
        with torch.no_grad():
            va_loss, va_acc, _ = run(val_dl, False)
        curves["train_acc"].append(tr_acc); curves["val_acc"].append(va_acc)
        curves["train_loss"].append(tr_loss); curves["val_loss"].append(va_loss)
        if verbose:
            log(f"Epoch {e + 1}/{num_epochs} | TrainAcc={tr_acc:.3f} ValAcc={va_acc:.3f}")
        if va_acc > best_val_acc:
            best_val_acc = va_acc
            if clf is not None:
                sd = clf.state_dict()
                best_state = sd.copy() if alias_best else {k: v.detach().clone() for k, v in sd.items()}
    if clf is not None and best_state is not None:
        clf.load_state_dict(best_state)
        clf.eval()
    with torch.no_grad():
        _, test_acc, _ = run(test_dl, False)
    return {"clf": clf, "best_val_acc": best_val_acc, "test_acc": test_acc, "best_state": best_state, **curves}


def grid_search_mlp(train_dl, val_dl, test_dl, lr_values=(1e-6, 5e-6, 1e-5, 5e-5, 1e-4, 5e-4, 1e-3, 5e-3, 1e-2, 5e-2, 1e-1),
                    input_dim=64, num_epochs=30, out_dir="mlp_best", device="cuda", verbose=True, log=print, fit_fn=None):
    """The reference's MLP learning-rate grid (R.md:2611-2732); saves `out_dir/MLP_GLOBAL_BEST.pt`."""
    os.makedirs(out_dir, exist_ok=True)
    fit_fn = fit_fn or fit_mlp
    best = {"val": 0, "test": 0, "lr": None, "state": None, "curves": None}
    for lr in lr_values:
        if verbose:
            log("\n=====================================")
            log(f"   Training MLP with LR = {lr}")
            log("=====================================")
        r = fit_fn(train_dl, val_dl, test_dl, lr, input_dim=input_dim, num_epochs=num_epochs, device=device, verbose=verbose, log=log)
        if r["best_val_acc"] > best["val"]:
            state = r["best_state"]
            best.update(val=r["best_val_acc"], test=r["test_acc"], lr=lr,
                        state=None if state is None else {k: v.detach().cpu().clone() for k, v in state.items()},
                        curves={k: r[k] for k in ("train_acc", "val_acc", "train_loss", "val_loss")})
    path = os.path.join(out_dir, "MLP_GLOBAL_BEST.pt")
    if best["state"] is not None:
        torch.save(best["state"], path)
    if verbose:
        log("\n--------------------------------------\nBest MLP\n--------------------------------------")
        log(f"Best LR             = {best['lr']}")
        log(f"Best Validation Acc = {best['val']:.4f}")
        log(f"Test Acc (finale)   = {best['test']:.4f}")
    return {"best_lr": best["lr"], "best_val_acc": best["val"], "test_acc": best["test"], "best_path": path, "curves": best["curves"]}


def evaluate(clf, test_dl):
    """Final evaluation of the reference (R.md:3171-3187): eval-mode argmax over a loader -> (preds, labels) numpy."""
    clf.eval()
    dev = next(clf.parameters()).device
    all_preds, all_labels = [], []
    with torch.no_grad():
        for xb, yb in test_dl:
            preds = clf(xb.to(dev)).argmax(1).cpu()
            all_preds.append(preds)
            all_labels.append(yb.cpu())
    return torch.cat(all_preds).numpy(), torch.cat(all_labels).numpy()
