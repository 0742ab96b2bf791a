"""GPU parity of the external MLP path (C ABI: eae_mlp_*) and of extract_features / evaluate / the fit loops.
The MLP kernels compute in fp32, so the tolerances against the reference goldens are fp32-level."""
import numpy as np
import pytest
import torch

import golden_util as gu
from helpers import mlp_state_np, ae_state_np, load_state_np

pytestmark = pytest.mark.gpu


def _clf(sd=None):
    import eae_amd
    torch.manual_seed(gu.MLP_SEED)
    c = eae_amd.MLP(input_dim=64, num_classes=10)
    load_state_np(c, sd if sd is not None else mlp_state_np())
    return c.to("cuda")


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_mlp_forward_and_grads(golden):
    from eae_amd.mlp_engine import mlp_engine_for
    g = golden("mlp_fwd_bwd_b64.npz")
    clf = _clf()
    clf.eval()
    with torch.no_grad():
        lg = clf(_cuda(g["x"]))
    assert np.abs(lg.cpu().numpy() - g["eval_logits"]).max() < 2e-5
    eng = mlp_engine_for(clf)
    clf.train()
    lg = eng.train_step(_cuda(g["x"]), _cuda(g["labels"]), lr=0.0, weight_decay=0.0, drop_mask=_cuda(g["drop_mask"]), want_logits=True)
    torch.cuda.synchronize()
    assert np.abs(lg.cpu().numpy() - g["logits"]).max() < 5e-5
    loss, acc, n = eng.read_stats()
    assert n == 64 and abs(loss - float(g["loss"])) < 1e-5
    for (p, i), (name, _) in zip(eng._slots, clf.named_parameters()):
        got = eng.grads[eng.poff[i]: eng.poff[i] + p.numel()].view(p.shape).cpu().numpy()
        ref = g[f"grad/{name}"]
        if np.abs(ref).max() < 1e-6:
            assert np.abs(got).max() < 1e-5, name
        else:
            assert np.abs(got - ref).max() <= 3e-4 * np.abs(ref).max(), name
    sd = clf.state_dict()
    for k in g.files:
        if k.startswith("buf/"):
            name = k[4:]
            if name.endswith("num_batches_tracked"):
                assert int(sd[name]) == int(g[k])
            else:
                np.testing.assert_allclose(sd[name].cpu().numpy(), g[k], rtol=1e-5, atol=1e-6)


def test_mlp_adam_trajectory(golden):
    from eae_amd.mlp_engine import mlp_engine_for
    g = golden("mlp_adam5.npz")
    clf = _clf()
    eng = mlp_engine_for(clf)
    clf.train()
    losses, correct = [], []
    for step, b in enumerate((64, 64, 64, 64, 48)):
        x, y = gu.make_latents(b, 400 + step)
        eng.reset_stats()
        eng.train_step(_cuda(x), _cuda(y), lr=float(g["lr"]), weight_decay=1e-4, drop_mask=_cuda(g[f"mask{step}"]))
        loss, acc, n = eng.read_stats()
        losses.append(loss)
        correct.append(int(round(acc * n)))
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=2e-4)
    assert correct == list(g["correct"])
    sd = clf.state_dict()
    for k in g.files:
        if k.startswith("final/"):
            name = k[6:]
            if name.endswith("num_batches_tracked"):
                assert int(sd[name]) == int(g[k])
            else:
                # Linear biases in front of BatchNorm1d (net.0.bias, net.4.bias) only see weight decay plus rounding
                # noise as gradient; Adam normalises that noise, so they agree to ~1e-4 absolute only
                atol = 3e-4 if name in ("net.0.bias", "net.4.bias") else 3e-5
                np.testing.assert_allclose(sd[name].cpu().numpy(), g[k], rtol=3e-3, atol=atol, err_msg=name)


def test_mlp_dropout_philox_rate():
    from eae_amd.mlp_engine import mlp_engine_for
    clf = _clf(mlp_state_np(perturb=False))
    eng = mlp_engine_for(clf)
    clf.train()
    x, _ = gu.make_latents(64, 7)
    with torch.no_grad():
        a = eng.forward(_cuda(x), train=True)
        b = eng.forward(_cuda(x), train=True)
    assert torch.equal(a, b)              # same (seed, step) -> same mask: counter-based RNG
    assert torch.isfinite(a).all()


def test_extract_features_and_module_forward(golden):
    import eae_amd
    g = golden("extract_features.npz")
    torch.manual_seed(gu.AE_SEED)
    m = eae_amd.SupervisedAutoencoder(64)
    load_state_np(m, ae_state_np())
    m = m.to("cuda")
    for p in m.enc.parameters():
        p.requires_grad = False          # R.md:2598-2599
    m.enc.eval()
    loader = []
    for i, b in enumerate((8, 5)):
        x, y = gu.make_images(b, 500 + i)
        loader.append((torch.from_numpy(x), torch.from_numpy(y)))
    X, Y = eae_amd.extract_features(loader, m.enc)
    assert X.device.type == "cpu" and X.shape == (13, 64) and Y.dtype == torch.int64
    assert np.abs(X.numpy() - g["X"]).max() <= 0.02 * np.abs(g["X"]).max()
    assert np.array_equal(Y.numpy(), g["y"])
    # module-level forward keeps the reference signatures under no_grad
    m.eval()
    with torch.no_grad():
        x_hat, logits, z = m(loader[0][0].cuda())
        x2 = m.dec(z)
    assert x_hat.shape == (8, 3, 64, 64) and logits.shape == (8, 10) and z.shape == (8, 64)
    assert np.abs(x2.cpu().numpy() - x_hat.cpu().numpy()).max() < 2e-2
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 32, 32, device="cuda"))      # wrong spatial size raises like the reference's Linear would


def test_fit_loops_end_to_end(tmp_path):
    """Tiny synthetic dataset through fit_autoencoder -> extract_features -> fit_mlp -> evaluate (R.md:599-3187)."""
    import eae_amd
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    n = 160
    y = rng.integers(0, 10, n)
    base = rng.random((10, 3, 64, 64)).astype(np.float32)
    x = np.clip(base[y] + 0.05 * rng.standard_normal((n, 3, 64, 64)).astype(np.float32), 0, 1)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y.astype(np.int64))

    def loader(lo, hi, bs=32):
        return [(xt[i:min(i + bs, hi)], yt[i:min(i + bs, hi)]) for i in range(lo, hi, bs)]

    tr, va, te = loader(0, 112), loader(112, 136), loader(136, 160)
    r = eae_amd.fit_autoencoder(tr, va, alpha=35, lr=5e-3, num_epochs=4, patience=15, verbose=False)
    assert r["epochs"] == 4 and np.isfinite(r["train_curve"]).all() and r["train_curve"][-1] < r["train_curve"][0]
    model = r["model"]
    torch.save(model.state_dict(), tmp_path / "AE.pt")
    m2 = eae_amd.SupervisedAutoencoder(64).to("cuda")
    m2.load_state_dict(torch.load(tmp_path / "AE.pt", weights_only=True))
    for p in m2.enc.parameters():
        p.requires_grad = False
    m2.enc.eval()
    Xtr, ytr = eae_amd.extract_features(tr, m2.enc)
    Xva, yva = eae_amd.extract_features(va, m2.enc)
    Xte, yte = eae_amd.extract_features(te, m2.enc)
    X1, _ = eae_amd.extract_features(tr, model.enc)
    assert torch.allclose(Xtr, X1)        # checkpoint round trip reproduces the encoder exactly

    def dl(X, Y, bs=64):
        return [(X[i:i + bs], Y[i:i + bs]) for i in range(0, len(X), bs)]

    r2 = eae_amd.fit_mlp(dl(Xtr, ytr), dl(Xva, yva), dl(Xte, yte), lr=1e-2, num_epochs=25, verbose=False)
    assert r2["train_acc"][-1] > 0.5 and 0.0 <= r2["test_acc"] <= 1.0
    preds, labels = eae_amd.evaluate(r2["clf"], dl(Xte, yte))
    assert preds.shape == labels.shape == (24,)
    assert abs((preds == labels).mean() - r2["test_acc"]) < 1e-6


def test_grid_search_drivers_on_the_engine(tmp_path):
    """N2 (R.md:599-729, 2611-2732): the grid drivers with their DEFAULT fit functions, i.e. on the HIP engine: every
    configuration gets a fresh model and optimizer state, the global best is tracked, AE_GLOBAL_BEST.pt /
    validation_losses.json / MLP_GLOBAL_BEST.pt are written with the reference's keys and load back into fresh modules."""
    import json
    import eae_amd
    torch.manual_seed(0)
    rng = np.random.default_rng(1)
    n = 96
    y = rng.integers(0, 10, n)
    base = rng.random((10, 3, 64, 64)).astype(np.float32)
    x = np.clip(base[y] + 0.05 * rng.standard_normal((n, 3, 64, 64)).astype(np.float32), 0, 1)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y.astype(np.int64))

    def loader(lo, hi, bs=32):
        return [(xt[i:min(i + bs, hi)], yt[i:min(i + bs, hi)]) for i in range(lo, hi, bs)]

    tr, va, te = loader(0, 64), loader(64, 80), loader(80, 96)
    logs = []
    g = eae_amd.grid_search_autoencoder(tr, va, alpha_values=(20, 35), lr_values=(1e-3, 5e-3), num_epochs=2, patience=15,
                                        out_dir=str(tmp_path / "models_best"), verbose=True, log=logs.append)
    assert set(g["results"].keys()) == {(20, 1e-3), (20, 5e-3), (35, 1e-3), (35, 5e-3)}
    assert all(np.isfinite(v) for v in g["results"].values())
    assert g["best_val_loss"] == min(g["results"].values()) and (g["best_alpha"], g["best_lr"]) == min(g["results"], key=g["results"].get)
    js = json.load(open(tmp_path / "models_best" / "validation_losses.json"))
    assert set(js.keys()) == {f"alpha={a}, lr={lr}" for a in (20, 35) for lr in (1e-3, 5e-3)}
    assert any(line.startswith("[AE α=35 LR=0.005] Epoch 2 | TrainLoss=") for line in logs)
    best_ae = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).to("cuda")
    best_ae.load_state_dict(torch.load(g["best_path"], weights_only=True))          # R.md:2594-2595
    for p_ in best_ae.enc.parameters():
        p_.requires_grad = False
    best_ae.enc.eval()
    feats = [eae_amd.extract_features(l, best_ae.enc) for l in (tr, va, te)]

    def dl(X, Y, bs=64):
        return [(X[i:i + bs], Y[i:i + bs]) for i in range(0, len(X), bs)]

    gm = eae_amd.grid_search_mlp(dl(*feats[0]), dl(*feats[1]), dl(*feats[2]), lr_values=(1e-3, 1e-2), num_epochs=4,
                                 out_dir=str(tmp_path / "mlp_best"), verbose=False)
    assert gm["best_lr"] in (1e-3, 1e-2) and 0.0 <= gm["test_acc"] <= 1.0 and len(gm["curves"]["val_acc"]) == 4
    clf = eae_amd.MLP(input_dim=64, num_classes=10).to("cuda")
    clf.load_state_dict(torch.load(gm["best_path"], weights_only=True))             # R.md:3172-3173
    preds, labels = eae_amd.evaluate(clf, dl(*feats[2]))
    assert preds.shape == labels.shape == (16,)


def test_mlp_autograd_drop_in_loop(golden):
    """The reference's MLP loop shape (R.md:2641-2646) with torch CE + torch.optim.Adam(weight_decay=1e-4) on the shell."""
    import torch.nn as nn
    from eae_amd.mlp_engine import mlp_engine_for
    g = golden("mlp_fwd_bwd_b64.npz")
    sd = mlp_state_np()
    clf = _clf(sd)
    clf.train()
    # make dropout a no-op for an exact comparison with the golden gradients: p is fixed at 0.3 in the kernel, so compare
    # against the engine's own fused step with the same Philox mask instead
    opt = torch.optim.Adam(clf.parameters(), lr=1e-3, weight_decay=1e-4)
    xb, yb = _cuda(g["x"]), _cuda(g["labels"])
    opt.zero_grad()
    logits = clf(xb)
    loss = nn.CrossEntropyLoss()(logits, yb)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in clf.named_parameters()}
    assert all(torch.isfinite(v).all() for v in grads.values())
    # finite-difference check of one weight through the same dropout mask (same seed/step -> same Philox stream)
    eng = mlp_engine_for(clf)
    w = clf.net[7].weight
    idx = (3, 5)
    eps = 1e-2
    from eae_amd.engine import _ptr, _stream
    from eae_amd._lib import check

    def loss_at(delta):
        with torch.no_grad():
            w[idx] += delta
            lg = torch.empty((64, 10), device="cuda")
            seed = (eng.seed + eng._autograd_calls) & 0xFFFFFFFFFFFFFFFF
            check(eng.lib.eae_mlp_forward(eng.ctx, _stream(), _ptr(xb), 64, 1, seed, None, _ptr(lg)))
            w[idx] -= delta
            return float(nn.CrossEntropyLoss()(lg, yb))

    fd = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
    assert abs(fd - float(grads["net.7.weight"][idx])) <= 2e-3 + 0.05 * abs(fd), (fd, float(grads["net.7.weight"][idx]))
    before = clf.net[0].weight.detach().clone()
    opt.step()
    assert not torch.equal(before, clf.net[0].weight)
