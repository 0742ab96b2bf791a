#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own classes.

Runs only in the build container (needs /root/reference).  It reads the notebook JSON,
execs the four model cells + extract_features (NB c31/c36/c40/c60/c63 -- SURVEY.md 8c) in a scratch namespace
holding only torch / nn, drives them with loops that follow R.md:642-658 and R.md:2639-2650 line by line,
and stores inputs + expected outputs as .npz.  No reference source text is written anywhere.

    python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_util as gu  # noqa: E402

NB = "/root/reference/Code/Hybrid_autoencoder–MLP_pipeline_for_satellite_image_classification.ipynb"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference():
    nb = json.load(open(NB, encoding="utf-8"))
    ns = {"torch": torch, "nn": nn, "device": torch.device("cpu")}
    for k in (31, 36, 40, 60, 63):
        exec("".join(nb["cells"][k]["source"]), ns)
    return ns


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def load_np(model, sd):
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})


def digest_sd(sd, prefix, store):
    for k, v in sd.items():
        d, s = gu.tensor_digest(v)
        store[f"{prefix}/{k}/digest"] = d
        store[f"{prefix}/{k}/sample"] = s


def gen_round2(ref):
    """Fixtures added in round 2 (`python tools/make_golden.py --round2` writes only these):
      ae_epoch.npz        one training epoch (batches 64, 64, 48) + one validation epoch (64, 56) driven line by line as
                          R.md:642-684: per-batch losses, the sample-weighted epoch means, eval-mode loss terms
      ae_adam1_joint_b8.npz  parameters after ONE Adam step (the first step moves every weight by -lr*sign(g): an elementwise
                          check of gradient signs + the optimizer kernel)"""
    SAE = ref["SupervisedAutoencoder"]

    def build_ae(latent=64):
        torch.manual_seed(gu.AE_SEED)
        m = SAE(latent_dim=latent, num_classes=10)
        load_np(m, gu.perturb_bn(sd_np(m)))
        return m

    alpha, lr = 35.0, 5e-3
    # ---- epoch accounting (R.md:638-684)
    m = build_ae()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    mse_fn, ce_fn = nn.MSELoss(), nn.CrossEntropyLoss()
    st = {"alpha": np.float32(alpha), "lr": np.float32(lr), "train_batches": np.array([64, 64, 48]), "val_batches": np.array([64, 56]),
          "train_seed0": np.int64(600), "val_seed0": np.int64(700)}
    m.train()
    train_loss, n_train = 0.0, 0
    tl, tm, tc = [], [], []
    for i, b in enumerate((64, 64, 48)):
        x, y = gu.make_images(b, 600 + i)
        imgs, labels = torch.from_numpy(x), torch.from_numpy(y)
        opt.zero_grad()
        x_hat, logits, _ = m(imgs)
        loss_recon = mse_fn(x_hat, imgs)
        loss_class = ce_fn(logits, labels)
        loss = alpha * loss_recon + loss_class
        loss.backward()
        opt.step()
        bs = imgs.size(0)
        train_loss += loss.item() * bs
        n_train += bs
        tl.append(loss.item()); tm.append(loss_recon.item()); tc.append(loss_class.item())
    train_loss /= n_train
    st.update(train_losses=np.array(tl, np.float32), train_mse=np.array(tm, np.float32), train_ce=np.array(tc, np.float32),
              train_epoch_loss=np.float32(train_loss), n_train=np.int64(n_train))
    m.eval()
    val_loss, n_val, vl, vm, vc, correct = 0.0, 0, [], [], [], 0
    with torch.no_grad():
        for i, b in enumerate((64, 56)):
            x, y = gu.make_images(b, 700 + i)
            imgs, labels = torch.from_numpy(x), torch.from_numpy(y)
            x_hat, logits, _ = m(imgs)
            loss_recon = mse_fn(x_hat, imgs)
            loss_class = ce_fn(logits, labels)
            loss = alpha * loss_recon + loss_class
            bs = imgs.size(0)
            val_loss += loss.item() * bs
            n_val += bs
            vl.append(loss.item()); vm.append(loss_recon.item()); vc.append(loss_class.item())
            correct += int((logits.argmax(1) == labels).sum())
    val_loss /= n_val
    st.update(val_losses=np.array(vl, np.float32), val_mse=np.array(vm, np.float32), val_ce=np.array(vc, np.float32),
              val_epoch_loss=np.float32(val_loss), n_val=np.int64(n_val), val_correct=np.int64(correct))
    np.savez_compressed(os.path.join(OUT, "ae_epoch.npz"), **st)

    # ---- one Adam step
    m = build_ae()
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    x, y = gu.make_images(8, 100)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    opt.zero_grad()
    xh, lg, _ = m(xt)
    loss = alpha * mse_fn(xh, xt) + ce_fn(lg, yt)
    loss.backward()
    opt.step()
    st = {"alpha": np.float32(alpha), "lr": np.float32(lr), "seed": np.int64(100), "loss": np.float32(loss.item())}
    digest_sd(sd_np(m), "final", st)
    np.savez_compressed(os.path.join(OUT, "ae_adam1_joint_b8.npz"), **st)
    for f in ("ae_epoch.npz", "ae_adam1_joint_b8.npz"):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1e6:.2f} MB")


def gen_latent48(ref):
    """`python tools/make_golden.py --latent48`: ae_latent48_b8.npz -- a latent width that is NOT a multiple of 64 (the reference
    takes any integer, R.md:309, 365, 423): train-mode forward, loss, gradient digests of every parameter, and the parameters
    after one Adam step, at B=8."""
    SAE = ref["SupervisedAutoencoder"]
    torch.manual_seed(gu.AE_SEED)
    m = SAE(latent_dim=48, num_classes=10)
    load_np(m, gu.perturb_bn(sd_np(m)))
    m.train()
    alpha, lr = 35.0, 5e-3
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    x, y = gu.make_images(8, 100)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    opt.zero_grad()
    xh, lg, z = m(xt)
    loss = alpha * nn.MSELoss()(xh, xt) + nn.CrossEntropyLoss()(lg, yt)
    loss.backward()
    st = {"alpha": np.float32(alpha), "lr": np.float32(lr), "seed": np.int64(100), "loss": np.float32(loss.item()),
          "x_hat": xh.detach().numpy().ravel()[::7].copy(), "logits": lg.detach().numpy(), "z": z.detach().numpy()}
    digest_sd({k: p.grad.detach().numpy() for k, p in m.named_parameters()}, "grad", st)
    opt.step()
    digest_sd(sd_np(m), "final", st)
    np.savez_compressed(os.path.join(OUT, "ae_latent48_b8.npz"), **st)
    print(f"  ae_latent48_b8.npz: {os.path.getsize(os.path.join(OUT, 'ae_latent48_b8.npz')) / 1e6:.2f} MB")


def gen_round3(ref):
    """`python tools/make_golden.py --round3`: ae_adam2_bn_b32.npz -- TWO Adam steps of the joint loss at B=32 (R.md:642-658) with every
    BatchNorm buffer recorded after each step.  Two steps in, the trajectory has not gone chaotic yet (the 5-step fixture has), so a
    wrong momentum, a biased-instead-of-unbiased running variance or a missed num_batches_tracked shows at a tight tolerance."""
    SAE = ref["SupervisedAutoencoder"]
    torch.manual_seed(gu.AE_SEED)
    m = SAE(latent_dim=64, num_classes=10)
    load_np(m, gu.perturb_bn(sd_np(m)))
    m.train()
    alpha, lr = 35.0, 5e-3
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    st = {"alpha": np.float32(alpha), "lr": np.float32(lr), "batch": np.int64(32), "seed0": np.int64(800)}
    losses = []
    for step in range(2):
        x, y = gu.make_images(32, 800 + step)
        xt, yt = torch.from_numpy(x), torch.from_numpy(y)
        opt.zero_grad()
        xh, lg, _ = m(xt)
        loss = alpha * nn.MSELoss()(xh, xt) + nn.CrossEntropyLoss()(lg, yt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                st[f"step{step + 1}/{k}"] = v.numpy().copy()
    st["losses"] = np.array(losses, np.float32)
    np.savez_compressed(os.path.join(OUT, "ae_adam2_bn_b32.npz"), **st)
    print(f"  ae_adam2_bn_b32.npz: {os.path.getsize(os.path.join(OUT, 'ae_adam2_bn_b32.npz')) / 1e6:.3f} MB")


def gen_round4(ref):
    """`python tools/make_golden.py --round4`: ae_nan_step_b8.npz -- what ONE step of the reference's loop (R.md:646-654) does to the
    model when the batch holds a non-finite value (x[3, 1, 10, 10] = inf, the input of tests/test_gpu_ae.py's divergence tests): the
    fraction of NaN entries of every parameter, of both Adam moments and of every BatchNorm buffer after the step, and the loss."""
    SAE = ref["SupervisedAutoencoder"]
    torch.manual_seed(gu.AE_SEED)
    m = SAE(latent_dim=64, num_classes=10)
    load_np(m, gu.perturb_bn(sd_np(m)))
    m.train()
    alpha, lr = 35.0, 5e-3
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    x, y = gu.make_images(8, 11)
    x = x.copy()
    x[3, 1, 10, 10] = np.inf
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    opt.zero_grad()
    xh, lg, _ = m(xt)
    loss = alpha * nn.MSELoss()(xh, xt) + nn.CrossEntropyLoss()(lg, yt)
    loss.backward()
    opt.step()
    st = {"alpha": np.float32(alpha), "lr": np.float32(lr), "seed": np.int64(11), "loss": np.float32(loss.item())}
    for k, v in m.state_dict().items():
        if v.is_floating_point():
            st[f"nan_frac/{k}"] = np.float32(torch.isnan(v).float().mean().item())
    for n, p in m.named_parameters():
        s_ = opt.state[p]
        st[f"nan_frac_m/{n}"] = np.float32(torch.isnan(s_["exp_avg"]).float().mean().item())
        st[f"nan_frac_v/{n}"] = np.float32(torch.isnan(s_["exp_avg_sq"]).float().mean().item())
    np.savez_compressed(os.path.join(OUT, "ae_nan_step_b8.npz"), **st)
    print("  ae_nan_step_b8.npz: loss", loss.item(), "param NaN fractions", sorted(set(float(v) for k, v in st.items() if k.startswith("nan_frac/") and "running" not in k)))
    print("  BatchNorm buffers:", {k.split("/", 1)[1]: float(v) for k, v in st.items() if k.startswith("nan_frac/") and "running" in k})


def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    if "--round4" in sys.argv:
        gen_round4(ref)
        return
    if "--round3" in sys.argv:
        gen_round3(ref)
        return
    if "--round2" in sys.argv:
        gen_round2(ref)
        return
    if "--latent48" in sys.argv:
        gen_latent48(ref)
        return
    SAE, MLP, Encoder, Decoder = ref["SupervisedAutoencoder"], ref["MLP"], ref["Encoder"], ref["Decoder"]

    # ---------------- init pin: same torch seed -> same default init
    st = {}
    for latent in (64, 128):
        torch.manual_seed(gu.AE_SEED)
        m = SAE(latent_dim=latent, num_classes=10)
        digest_sd(sd_np(m), f"ae{latent}", st)
    torch.manual_seed(gu.MLP_SEED)
    digest_sd(sd_np(MLP(input_dim=64, num_classes=10)), "mlp64", st)
    np.savez_compressed(os.path.join(OUT, "init_digest.npz"), **st)

    # ---------------- G1/G2/G3: forward (train+eval), grads, intermediates
    def build_ae(latent=64):
        torch.manual_seed(gu.AE_SEED)
        m = SAE(latent_dim=latent, num_classes=10)
        sd = gu.perturb_bn(sd_np(m))
        load_np(m, sd)
        return m

    alpha = 35.0
    for b, seed, full_grads in ((8, 100, True), (2, 101, False), (32, 102, False), (48, 103, False), (56, 104, False)):
        m = build_ae()
        x, y = gu.make_images(b, seed)
        xt, yt = torch.from_numpy(x), torch.from_numpy(y)
        small = b <= 8      # big batches: inputs are regenerated from the seed, images stored as strided samples
        st = {"labels": y, "alpha": np.float32(alpha), "seed": np.int64(seed)}
        if small:
            st["x"] = x
        # eval forward first (does not mutate buffers)
        m.eval()
        with torch.no_grad():
            xh, lg, z = m(xt)
        st.update(eval_logits=lg.numpy(), eval_z=z.numpy())
        st["eval_x_hat"] = xh.numpy() if small else xh.numpy().ravel()[::7].copy()
        # train forward + loss + backward   (R.md:646-653)
        m.train()
        inter = {}
        if b == 2:
            hooks = []
            for name, mod in list(m.enc.encoder.named_children()) + list(m.dec.decoder.named_children()):
                pass
            for idx in (0, 2, 3, 5, 9, 11):
                hooks.append(m.enc.encoder[idx].register_forward_hook(
                    lambda mod, i, o, idx=idx: inter.__setitem__(f"enc.encoder.{idx}", o.detach().numpy().copy())))
            for idx in (1, 3, 4, 6, 7, 9, 10):
                hooks.append(m.dec.decoder[idx].register_forward_hook(
                    lambda mod, i, o, idx=idx: inter.__setitem__(f"dec.decoder.{idx}", o.detach().numpy().copy())))
            hooks.append(m.dec.decoder_input.register_forward_hook(
                lambda mod, i, o: inter.__setitem__("dec.decoder_input", o.detach().numpy().copy())))
        m.zero_grad()
        xh, lg, z = m(xt)
        z.retain_grad()
        l_r = nn.MSELoss()(xh, xt)
        l_c = nn.CrossEntropyLoss()(lg, yt)
        loss = alpha * l_r + l_c
        loss.backward()
        st["x_hat"] = xh.detach().numpy() if small else xh.detach().numpy().ravel()[::7].copy()
        st.update(logits=lg.detach().numpy(), z=z.detach().numpy(),
                  loss=np.float32(loss.item()), loss_recon=np.float32(l_r.item()), loss_class=np.float32(l_c.item()),
                  dz=z.grad.numpy())
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                st[f"buf/{k}"] = v.numpy().copy()
        for k, p in m.named_parameters():
            if full_grads:
                st[f"grad/{k}"] = p.grad.numpy().copy()
            else:
                d, s = gu.tensor_digest(p.grad.numpy())
                st[f"gradd/{k}/digest"] = d
                st[f"gradd/{k}/sample"] = s
        for k, v in inter.items():
            st[f"inter/{k}"] = v
        np.savez_compressed(os.path.join(OUT, f"ae_fwd_bwd_b{b}.npz"), **st)

    # ---------------- G8: b=1 eval
    m = build_ae()
    m.eval()
    x, y = gu.make_images(1, 105)
    with torch.no_grad():
        xh, lg, z = m(torch.from_numpy(x))
    np.savez_compressed(os.path.join(OUT, "ae_eval_b1.npz"), x=x, labels=y, eval_x_hat=xh.numpy(),
                        eval_logits=lg.numpy(), eval_z=z.numpy())

    # ---------------- G9: latent 128 forward in train mode under no_grad (probe cell c44, R.md:504-513)
    m = build_ae(128)
    m.train()
    x, y = gu.make_images(8, 106)
    with torch.no_grad():
        xh, lg, z = m(torch.from_numpy(x))
        ce = nn.CrossEntropyLoss()(lg, torch.from_numpy(y)).item()
        ms = nn.MSELoss()(xh, torch.from_numpy(x)).item()
    np.savez_compressed(os.path.join(OUT, "ae_latent128_b8.npz"), x=x, labels=y, x_hat=xh.numpy(), logits=lg.numpy(),
                        z=z.numpy(), ce=np.float32(ce), mse=np.float32(ms))

    # ---------------- G4: 5 Adam steps, joint loss (R.md:642-658)
    for tag, head in (("joint", True), ("recon", False)):
        m = build_ae()
        m.train()
        params = list(m.parameters()) if head else list(m.enc.parameters()) + list(m.dec.parameters())
        opt = torch.optim.Adam(params, lr=5e-3)
        st = {"alpha": np.float32(alpha), "lr": np.float32(5e-3)}
        losses = []
        for step in range(5):
            x, y = gu.make_images(8, 200 + step)
            xt, yt = torch.from_numpy(x), torch.from_numpy(y)
            opt.zero_grad()
            if head:
                xh, lg, _ = m(xt)
                loss = alpha * nn.MSELoss()(xh, xt) + nn.CrossEntropyLoss()(lg, yt)
            else:   # config c2: encoder+decoder reconstruction (MSE) only
                xh = m.dec(m.enc(xt))
                loss = nn.MSELoss()(xh, xt)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        st["losses"] = np.array(losses, np.float32)
        digest_sd(sd_np(m), "final", st)
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                st[f"buf/{k}"] = v.numpy().copy()
        np.savez_compressed(os.path.join(OUT, f"ae_adam5_{tag}_b8.npz"), **st)

    # ---------------- G6: MLP
    def build_mlp():
        torch.manual_seed(gu.MLP_SEED)
        c = MLP(input_dim=64, num_classes=10)
        load_np(c, gu.perturb_bn(sd_np(c), seed=9))
        return c

    c = build_mlp()
    x, y = gu.make_latents(64, 300)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    st = {"x": x, "labels": y}
    c.eval()
    with torch.no_grad():
        st["eval_logits"] = c(xt).numpy()
    c.train()
    masks = []
    hk = c.net[3].register_forward_hook(
        lambda mod, i, o: masks.append(((o != 0) | (i[0] == 0)).numpy().astype(np.float32)))
    torch.manual_seed(1234)
    c.zero_grad()
    lg = c(xt)
    loss = nn.CrossEntropyLoss()(lg, yt)
    loss.backward()
    st.update(logits=lg.detach().numpy(), loss=np.float32(loss.item()), drop_mask=masks[-1])
    for k, p in c.named_parameters():
        st[f"grad/{k}"] = p.grad.numpy().copy()
    for k, v in c.state_dict().items():
        if "running" in k or "num_batches" in k:
            st[f"buf/{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "mlp_fwd_bwd_b64.npz"), **st)

    # 5 Adam(wd=1e-4) steps  (R.md:2625, 2639-2646); batches of 64, 64, 64, 64, 48
    c = build_mlp()
    c.train()
    masks.clear()
    hk = c.net[3].register_forward_hook(
        lambda mod, i, o: masks.append(((o != 0) | (i[0] == 0)).numpy().astype(np.float32)))
    opt = torch.optim.Adam(c.parameters(), lr=1e-3, weight_decay=1e-4)
    st = {"lr": np.float32(1e-3)}
    losses, correct = [], []
    torch.manual_seed(4321)
    for step, b in enumerate((64, 64, 64, 64, 48)):
        x, y = gu.make_latents(b, 400 + step)
        xt, yt = torch.from_numpy(x), torch.from_numpy(y)
        opt.zero_grad()
        lg = c(xt)
        loss = nn.CrossEntropyLoss()(lg, yt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        correct.append((lg.argmax(1) == yt).sum().item())
        st[f"mask{step}"] = masks[-1]
    st["losses"] = np.array(losses, np.float32)
    st["correct"] = np.array(correct, np.int64)
    for k, v in c.state_dict().items():
        st[f"final/{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "mlp_adam5.npz"), **st)

    # ---------------- G7: extract_features on a 2-batch synthetic loader (R.md:2498-2510)
    m = build_ae()
    loader = []
    for i, b in enumerate((8, 5)):
        x, y = gu.make_images(b, 500 + i)
        loader.append((torch.from_numpy(x), torch.from_numpy(y)))
    X, Y = ref["extract_features"](loader, m.enc)
    np.savez_compressed(os.path.join(OUT, "extract_features.npz"), X=X.numpy(), y=Y.numpy())
    gen_round2(ref)
    gen_latent48(ref)
    gen_round3(ref)
    print("fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
