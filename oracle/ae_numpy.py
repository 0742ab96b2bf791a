"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the conv-autoencoder + MLP training path.

This file is a from-the-formulas NumPy restatement of the arithmetic that the reference
notebook reaches through ``torch.nn`` on its hot path.  It is *not* part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker.  The product path (``csrc/`` HIP kernels behind the
C ABI in ``include/eae.h``) never routes through this module.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py`` against the
fixtures under ``tests/golden/`` which were produced by importing the reference's own
classes from the notebook JSON (``tools/make_golden.py``; the reference ships no tests or
golden vectors of its own).

Reference citations use ``R.md:N`` = line N of
``/root/reference/Report/Hybrid_autoencoder–MLP_pipeline_for_satellite_image_classification.md``.

Layouts follow the reference (NCHW activations, ``[Cout,Cin,3,3]`` conv weights,
``[Cin,Cout,3,3]`` transposed-conv weights, state-dict key names of R.md:287-433 and
R.md:2549-2566).  ``quant="bf16"`` mirrors the rounding points of the HIP path (bf16
storage of weights/activations/gradients, fp32 accumulation and statistics) so that the
kernels can be checked tightly; ``quant=None`` is the reference's pure-fp32 arithmetic.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

BN_EPS = 1e-5          # torch.nn.BatchNorm default (R.md:293 uses defaults)
BN_MOMENTUM = 0.1

ENC_CONVS = ((0, 1), (3, 4), (6, 7), (9, 10))     # (conv idx, bn idx) in enc.encoder  R.md:291-306
DEC_DECONVS = ((1, 2), (4, 5), (7, 8), (10, None))  # (deconv idx, bn idx) in dec.decoder R.md:367-384


# --------------------------------------------------------------------------------------
# rounding helpers
# --------------------------------------------------------------------------------------
def bf16_round(a):
    """Round fp32 -> bf16 (round-to-nearest-even) and return as fp32."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)
    return out.reshape(a.shape)


def _q(a, quant):
    return bf16_round(a) if quant in ("bf16", "fp8") else a


def fp8_round(a, fmt):
    """Round to OCP fp8 (round-to-nearest-even, saturating, with subnormals) and return as fp32.  fmt = "e4m3" (max 448, 3
    mantissa bits, min normal 2^-6) or "e5m2" (max 57344, 2 mantissa bits, min normal 2^-14): what v_cvt_scalef32_pk_{fp8,bf8}_bf16
    produces on gfx950 with MODE.FP16_OVFL set (tools/probe/probe_fp8.hip)."""
    mant, emin, maxv = (3, -6, 448.0) if fmt == "e4m3" else (2, -14, 57344.0)
    a = np.asarray(a, np.float32)
    m = np.minimum(np.abs(a).astype(np.float64), maxv)
    e = np.maximum(np.floor(np.log2(np.maximum(m, 1e-300))), emin)
    q = np.exp2(e - mant)
    r = np.minimum(np.round(m / q) * q, maxv)          # np.round: half to even
    return (np.sign(a) * r).astype(np.float32)


def fp8_bytes_e4m3(a):
    """OCP e4m3 byte encoding of values that are already exactly representable (the output of fp8_round(., "e4m3"))."""
    a = np.asarray(a, np.float32)
    m = np.abs(a).astype(np.float64)
    sign = (np.signbit(a)).astype(np.uint8) << 7
    e = np.floor(np.log2(np.maximum(m, 2.0 ** -20)))
    normal = m >= 2.0 ** -6
    eb = np.where(normal, e + 7, 0).astype(np.int64)
    mant = np.where(normal, np.round((m / np.exp2(e) - 1.0) * 8), np.round(m * 512)).astype(np.int64)
    return (sign | (eb.astype(np.uint8) << 3) | mant.astype(np.uint8)).astype(np.uint8)


def _q8(a, scale, fmt):
    """Operand of an fp8 GEMM: the (already bf16-rounded) value times its power-of-two scale, rounded to fp8, divided back."""
    sc = np.float32(scale)
    return fp8_round(a * sc, fmt) / sc


# 3x3 layers whose GEMMs take fp8 operands in the fp8 variant, in the engine's order (include/eae.h: eae_fp8_scales)
FP8_ENC = {1: 0, 2: 1, 3: 2}      # ENC_CONVS index -> scale slot (conv2, conv3, conv4)
FP8_DEC = {0: 3, 1: 4, 2: 5}      # DEC_DECONVS index -> scale slot (deconv1, deconv2, deconv3)


# --------------------------------------------------------------------------------------
# primitive ops (forward + backward), NCHW fp32
# --------------------------------------------------------------------------------------
def conv_s2_fwd(x, w, b):
    """3x3 stride-2 pad-1 convolution (nn.Conv2d(...,3,stride=2,padding=1), R.md:292).

    y[n,co,oy,ox] = b[co] + sum_{ci,ky,kx} x[n,ci,2oy-1+ky,2ox-1+kx] * w[co,ci,ky,kx]
    """
    n, ci, h, wd = x.shape
    ho, wo = h // 2, wd // 2
    xp = np.zeros((n, ci, h + 2, wd + 2), np.float32)
    xp[:, :, 1:h + 1, 1:wd + 1] = x
    y = np.zeros((n, w.shape[0], ho, wo), np.float32)
    for ky in range(3):
        for kx in range(3):
            patch = xp[:, :, ky:ky + 2 * ho:2, kx:kx + 2 * wo:2]
            y += np.einsum("nchw,oc->nohw", patch, w[:, :, ky, kx], optimize=True)
    if b is not None:
        y += b[None, :, None, None]
    return y


def conv_s2_bwd(x, w, dy):
    """Gradients of conv_s2_fwd w.r.t. x, w, b."""
    n, ci, h, wd = x.shape
    ho, wo = h // 2, wd // 2
    xp = np.zeros((n, ci, h + 2, wd + 2), np.float32)
    xp[:, :, 1:h + 1, 1:wd + 1] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for ky in range(3):
        for kx in range(3):
            patch = xp[:, :, ky:ky + 2 * ho:2, kx:kx + 2 * wo:2]
            dw[:, :, ky, kx] = np.einsum("nohw,nchw->oc", dy, patch, optimize=True)
            dxp[:, :, ky:ky + 2 * ho:2, kx:kx + 2 * wo:2] += np.einsum(
                "nohw,oc->nchw", dy, w[:, :, ky, kx], optimize=True)
    return dxp[:, :, 1:h + 1, 1:wd + 1], dw, dy.sum(axis=(0, 2, 3))


def deconv_s2_fwd(x, w, b):
    """3x3 stride-2 pad-1 output_padding-1 transposed conv (nn.ConvTranspose2d, R.md:370).

    y[n,co,oy,ox] = b[co] + sum_{ci,ky,kx : oy = 2iy-1+ky, ox = 2ix-1+kx} x[n,ci,iy,ix] * w[ci,co,ky,kx]
    Output is exactly 2H x 2W.
    """
    n, ci, h, wd = x.shape
    co = w.shape[1]
    yp = np.zeros((n, co, 2 * h + 2, 2 * wd + 2), np.float32)   # index = oy + 1
    for ky in range(3):
        for kx in range(3):
            yp[:, :, ky:ky + 2 * h:2, kx:kx + 2 * wd:2] += np.einsum(
                "nchw,co->nohw", x, w[:, :, ky, kx], optimize=True)
    y = yp[:, :, 1:2 * h + 1, 1:2 * wd + 1].copy()
    if b is not None:
        y += b[None, :, None, None]
    return y


def deconv_s2_bwd(x, w, dy):
    n, ci, h, wd = x.shape
    co = w.shape[1]
    dyp = np.zeros((n, co, 2 * h + 2, 2 * wd + 2), np.float32)
    dyp[:, :, 1:2 * h + 1, 1:2 * wd + 1] = dy
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for ky in range(3):
        for kx in range(3):
            g = dyp[:, :, ky:ky + 2 * h:2, kx:kx + 2 * wd:2]
            dx += np.einsum("nohw,co->nchw", g, w[:, :, ky, kx], optimize=True)
            dw[:, :, ky, kx] = np.einsum("nchw,nohw->co", x, g, optimize=True)
    return dx, dw, dy.sum(axis=(0, 2, 3))


def bn_fwd(y, gamma, beta, rmean, rvar, train, axes):
    """BatchNorm forward (nn.BatchNorm2d / BatchNorm1d defaults).

    Train: normalise with the biased batch variance, update running stats with the unbiased one.
    Returns (out, cache, new_rmean, new_rvar).
    """
    shape = [1] * y.ndim
    shape[1] = -1
    if train:
        cnt = y.size // y.shape[1]
        mean = y.mean(axis=axes, dtype=np.float64).astype(np.float32)
        var = y.var(axis=axes, dtype=np.float64).astype(np.float32)          # biased
        unb = var * (cnt / max(cnt - 1, 1))
        new_rm = (1 - BN_MOMENTUM) * rmean + BN_MOMENTUM * mean
        new_rv = (1 - BN_MOMENTUM) * rvar + BN_MOMENTUM * unb
    else:
        mean, var = rmean, rvar
        new_rm, new_rv = rmean, rvar
    invstd = (1.0 / np.sqrt(var.astype(np.float32) + np.float32(BN_EPS))).astype(np.float32)
    xhat = (y - mean.reshape(shape)) * invstd.reshape(shape)
    out = gamma.reshape(shape) * xhat + beta.reshape(shape)
    return out.astype(np.float32), (xhat.astype(np.float32), invstd), new_rm.astype(np.float32), new_rv.astype(np.float32)


def bn_bwd(dout, gamma, cache, axes):
    """BatchNorm training-mode backward (native_batch_norm_backward)."""
    xhat, invstd = cache
    shape = [1] * dout.ndim
    shape[1] = -1
    cnt = dout.size // dout.shape[1]
    dgamma = (dout * xhat).sum(axis=axes, dtype=np.float64).astype(np.float32)
    dbeta = dout.sum(axis=axes, dtype=np.float64).astype(np.float32)
    dx = (gamma * invstd).reshape(shape) / cnt * (cnt * dout - dbeta.reshape(shape) - xhat * dgamma.reshape(shape))
    return dx.astype(np.float32), dgamma, dbeta


def linear_fwd(x, w, b):
    return x @ w.T + b


def linear_bwd(x, w, dy):
    return dy @ w, dy.T @ x, dy.sum(axis=0)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


def log_softmax(x):
    x = x.astype(np.float64)
    m = x.max(axis=1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=1, keepdims=True))


def cross_entropy(logits, labels):
    """nn.CrossEntropyLoss() mean reduction (R.md:623). Returns (loss, dlogits)."""
    ls = log_softmax(logits)
    b = logits.shape[0]
    loss = -ls[np.arange(b), labels].mean()
    d = np.exp(ls)
    d[np.arange(b), labels] -= 1.0
    return np.float32(loss), (d / b).astype(np.float32)


def mse(x_hat, x):
    """nn.MSELoss() mean over all elements (R.md:622). Returns (loss, d x_hat)."""
    diff = x_hat.astype(np.float64) - x.astype(np.float64)
    return np.float32((diff * diff).mean()), (2.0 * diff / diff.size).astype(np.float32)


# --------------------------------------------------------------------------------------
# SupervisedAutoencoder (R.md:416-433): forward, loss, backward
# --------------------------------------------------------------------------------------
def ae_forward(p, x, train, quant=None, head=True, scales=None):
    """Forward of SupervisedAutoencoder.forward (R.md:429-433) -> dict with x_hat, logits, z.

    ``p`` maps reference state-dict names to fp32 arrays.  Returns also ``cache`` for backward and
    ``new_buffers`` (running stats after the call, train mode only).
    """
    cache = {"x": x}
    newb = OrderedDict()
    a = _q(x, quant)
    # ---- Encoder R.md:291-310
    for li, (ci_, bi_) in enumerate(ENC_CONVS):
        w = _q(p[f"enc.encoder.{ci_}.weight"], quant)
        if quant == "fp8" and li in FP8_ENC:      # quant="fp8": `scales` = {"act": [6], "grad": [6], "w": [6]} (the engine's, delayed)
            j = FP8_ENC[li]
            y = conv_s2_fwd(_q8(a, scales["act"][j], "e4m3"), _q8(w, scales["w"][j], "e4m3"), p[f"enc.encoder.{ci_}.bias"])
        else:
            y = conv_s2_fwd(a, w, p[f"enc.encoder.{ci_}.bias"])
        y = _q(y, quant)                        # raw conv output is stored bf16 on the HIP path
        o, bc, rm, rv = bn_fwd(y, p[f"enc.encoder.{bi_}.weight"], p[f"enc.encoder.{bi_}.bias"],
                               p[f"enc.encoder.{bi_}.running_mean"], p[f"enc.encoder.{bi_}.running_var"],
                               train, (0, 2, 3))
        if train:
            newb[f"enc.encoder.{bi_}.running_mean"] = rm
            newb[f"enc.encoder.{bi_}.running_var"] = rv
        act = np.maximum(o, 0.0)
        cache[f"enc{li}"] = (a, y, bc, o)
        a = _q(act, quant)
    n = x.shape[0]
    flat = a.reshape(n, -1)                     # nn.Flatten on NCHW: index c*16+h*4+w  (R.md:308)
    z = linear_fwd(flat, _q(p["enc.encoder.13.weight"], quant), p["enc.encoder.13.bias"])
    cache["flat"] = flat
    cache["z"] = z
    # ---- Decoder R.md:365-384
    d = linear_fwd(_q(z, quant), _q(p["dec.decoder_input.weight"], quant), p["dec.decoder_input.bias"])
    d = _q(d, quant)
    a = d.reshape(n, 256, x.shape[2] // 16, x.shape[3] // 16)    # nn.Unflatten(1,(256,4,4))
    for li, (di_, bi_) in enumerate(DEC_DECONVS):
        w = _q(p[f"dec.decoder.{di_}.weight"], quant)
        if quant == "fp8" and li in FP8_DEC:
            j = FP8_DEC[li]
            y = deconv_s2_fwd(_q8(a, scales["act"][j], "e4m3"), _q8(w, scales["w"][j], "e4m3"), p[f"dec.decoder.{di_}.bias"])
        else:
            y = deconv_s2_fwd(a, w, p[f"dec.decoder.{di_}.bias"])
        if bi_ is None:
            x_hat = sigmoid(y)                  # nn.Sigmoid R.md:383
            cache[f"dec{li}"] = (a, y, None, None)
            break
        y = _q(y, quant)
        o, bc, rm, rv = bn_fwd(y, p[f"dec.decoder.{bi_}.weight"], p[f"dec.decoder.{bi_}.bias"],
                               p[f"dec.decoder.{bi_}.running_mean"], p[f"dec.decoder.{bi_}.running_var"],
                               train, (0, 2, 3))
        if train:
            newb[f"dec.decoder.{bi_}.running_mean"] = rm
            newb[f"dec.decoder.{bi_}.running_var"] = rv
        cache[f"dec{li}"] = (a, y, bc, o)
        a = _q(np.maximum(o, 0.0), quant)
    out = {"x_hat": x_hat, "z": z, "cache": cache, "new_buffers": newb}
    # ---- classifier R.md:423-427
    if head:
        h_pre = linear_fwd(z, p["classifier.0.weight"], p["classifier.0.bias"])
        h = np.maximum(h_pre, 0.0)
        logits = linear_fwd(h, p["classifier.2.weight"], p["classifier.2.bias"])
        cache["h_pre"] = h_pre
        cache["h"] = h
        out["logits"] = logits
    return out


def ae_loss(out, x, labels, alpha, head=True):
    """loss = alpha * MSE(x_hat, x) + CE(logits, labels)   (R.md:649-651)."""
    l_r, _ = mse(out["x_hat"], x)
    if head:
        l_c, _ = cross_entropy(out["logits"], labels)
    else:
        l_c = np.float32(0.0)
    return np.float32(alpha * l_r + l_c), l_r, l_c


def _bwd8(fn, a, w, dy, j, scales):
    """Backward of one 3x3 layer with fp8 GEMM operands: backward-data from (w e4m3, dy e5m2), weight gradient from (a e4m3, dy e5m2)."""
    dy8 = _q8(dy, scales["grad"][j], "e5m2")
    da, _, db = fn(a, _q8(w, scales["w"][j], "e4m3"), dy8)
    _, dw, _ = fn(_q8(a, scales["act"][j], "e4m3"), w, dy8)
    return da, dw, db


def ae_backward(p, out, x, labels, alpha, quant=None, head=True, dout=None, scales=None):
    """Gradients of alpha*MSE + CE w.r.t. all 38 parameters (autograd of R.md:653) + dz.

    ``dout`` = optional externally supplied (dx_hat, dlogits, dz) replacing the fused loss.
    """
    c = out["cache"]
    g = OrderedDict()
    n = x.shape[0]
    if dout is None:
        _, dxh = mse(out["x_hat"], x)
        dxh = dxh * np.float32(alpha)
        dz_extra = None
        if head:
            _, dlog = cross_entropy(out["logits"], labels)
    else:
        dxh, dlog, dz_extra = dout
    # classifier
    dz = np.zeros_like(out["z"])
    if head and dlog is not None:
        dh, g["classifier.2.weight"], g["classifier.2.bias"] = linear_bwd(c["h"], p["classifier.2.weight"], dlog)
        dh = dh * (c["h_pre"] > 0)
        dzc, g["classifier.0.weight"], g["classifier.0.bias"] = linear_bwd(c["z"], p["classifier.0.weight"], dh)
        dz = dz + dzc
    if dz_extra is not None:
        dz = dz + dz_extra
    # decoder, last layer: sigmoid
    a, y, _, _ = c["dec3"]
    xh = out["x_hat"]
    dy = _q(dxh * xh * (1.0 - xh), quant)
    da, dw, db = deconv_s2_bwd(a, _q(p["dec.decoder.10.weight"], quant), dy)
    g["dec.decoder.10.weight"], g["dec.decoder.10.bias"] = dw, db
    for li in (2, 1, 0):
        di_, bi_ = DEC_DECONVS[li]
        a, y, bc, o = c[f"dec{li}"]
        gm = _q(da * (o > 0), quant)                       # ReLU mask; stored bf16 on the HIP path
        dy, dgam, dbet = bn_bwd(gm, p[f"dec.decoder.{bi_}.weight"], bc, (0, 2, 3))
        g[f"dec.decoder.{bi_}.weight"], g[f"dec.decoder.{bi_}.bias"] = dgam, dbet
        dy = _q(dy, quant)
        if quant == "fp8":
            da, dw, db = _bwd8(deconv_s2_bwd, a, _q(p[f"dec.decoder.{di_}.weight"], quant), dy, FP8_DEC[li], scales)
        else:
            da, dw, db = deconv_s2_bwd(a, _q(p[f"dec.decoder.{di_}.weight"], quant), dy)
        g[f"dec.decoder.{di_}.weight"], g[f"dec.decoder.{di_}.bias"] = dw, db
    dd = _q(da.reshape(n, -1), quant)
    dzd, g["dec.decoder_input.weight"], g["dec.decoder_input.bias"] = linear_bwd(
        _q(c["z"], quant), _q(p["dec.decoder_input.weight"], quant), dd)
    dz = dz + dzd
    g["dz"] = dz
    # encoder
    dflat, g["enc.encoder.13.weight"], g["enc.encoder.13.bias"] = linear_bwd(
        c["flat"], _q(p["enc.encoder.13.weight"], quant), _q(dz, quant))
    da = dflat.reshape(c["enc3"][3].shape)
    for li in (3, 2, 1, 0):
        ci_, bi_ = ENC_CONVS[li]
        a, y, bc, o = c[f"enc{li}"]
        gm = _q(da * (o > 0), quant)
        dy, dgam, dbet = bn_bwd(gm, p[f"enc.encoder.{bi_}.weight"], bc, (0, 2, 3))
        g[f"enc.encoder.{bi_}.weight"], g[f"enc.encoder.{bi_}.bias"] = dgam, dbet
        dy = _q(dy, quant)
        if quant == "fp8" and li in FP8_ENC:
            da, dw, db = _bwd8(conv_s2_bwd, a, _q(p[f"enc.encoder.{ci_}.weight"], quant), dy, FP8_ENC[li], scales)
        else:
            da, dw, db = conv_s2_bwd(a, _q(p[f"enc.encoder.{ci_}.weight"], quant), dy)
        g[f"enc.encoder.{ci_}.weight"], g[f"enc.encoder.{ci_}.bias"] = dw, db
    g["dx"] = da
    return g


# --------------------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults, R.md:624; MLP adds weight_decay=1e-4, R.md:2625)
# --------------------------------------------------------------------------------------
def adam_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One Adam step in place on ``params`` (dict name -> fp32 array).

    state: {"step": int, "m": {...}, "v": {...}}.  L2 weight decay is added to the gradient
    (coupled, not AdamW), bias-corrected as torch does: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
    """
    state["step"] += 1
    t = state["step"]
    b1, b2 = betas
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k, gk in grads.items():
        if k not in params or k in ("dz", "dx"):
            continue
        pk = params[k]
        gk = gk.astype(np.float32)
        if weight_decay:
            gk = gk + np.float32(weight_decay) * pk
        m = state["m"].setdefault(k, np.zeros_like(pk))
        v = state["v"].setdefault(k, np.zeros_like(pk))
        m *= np.float32(b1)
        m += np.float32(1 - b1) * gk
        v *= np.float32(b2)
        v += np.float32(1 - b2) * gk * gk
        denom = np.sqrt(v) / np.float32(bc2_sqrt) + np.float32(eps)
        pk -= np.float32(step_size) * (m / denom)


def ae_train_step(p, state, x, labels, alpha, lr, quant=None, head=True):
    """One iteration of the reference's AE batch loop (R.md:646-654). Mutates p and state."""
    out = ae_forward(p, x, True, quant=quant, head=head)
    loss, l_r, l_c = ae_loss(out, x, labels, alpha, head=head)
    g = ae_backward(p, out, x, labels, alpha, quant=quant, head=head)
    for k, v in out["new_buffers"].items():
        p[k] = v
    for k in list(p.keys()):
        if k.endswith("num_batches_tracked"):
            p[k] = p[k] + 1
    adam_step(p, g, state, lr)
    return loss, l_r, l_c, out, g


# --------------------------------------------------------------------------------------
# external MLP (R.md:2549-2566): Linear-BN1d-ReLU-Dropout(0.3)-Linear-BN1d-ReLU-Linear
# --------------------------------------------------------------------------------------
def mlp_forward(p, x, train, drop_mask=None, p_drop=0.3):
    """MLP.forward (R.md:2565). ``drop_mask`` = Bernoulli keep mask [B,128] (train mode); None = no dropout."""
    c = {"x": x}
    newb = OrderedDict()
    h1 = linear_fwd(x, p["net.0.weight"], p["net.0.bias"])
    o1, bc1, rm, rv = bn_fwd(h1, p["net.1.weight"], p["net.1.bias"], p["net.1.running_mean"],
                             p["net.1.running_var"], train, (0,))
    if train:
        newb["net.1.running_mean"], newb["net.1.running_var"] = rm, rv
    a1 = np.maximum(o1, 0.0)
    if train and drop_mask is not None:
        a1d = a1 * drop_mask / np.float32(1.0 - p_drop)
    else:
        a1d = a1
    h2 = linear_fwd(a1d, p["net.4.weight"], p["net.4.bias"])
    o2, bc2, rm, rv = bn_fwd(h2, p["net.5.weight"], p["net.5.bias"], p["net.5.running_mean"],
                             p["net.5.running_var"], train, (0,))
    if train:
        newb["net.5.running_mean"], newb["net.5.running_var"] = rm, rv
    a2 = np.maximum(o2, 0.0)
    logits = linear_fwd(a2, p["net.7.weight"], p["net.7.bias"])
    c.update(o1=o1, bc1=bc1, a1d=a1d, o2=o2, bc2=bc2, a2=a2, drop_mask=drop_mask, p_drop=p_drop, train=train)
    return {"logits": logits.astype(np.float32), "cache": c, "new_buffers": newb}


def mlp_backward(p, out, labels):
    c = out["cache"]
    g = OrderedDict()
    loss, dlog = cross_entropy(out["logits"], labels)
    da2, g["net.7.weight"], g["net.7.bias"] = linear_bwd(c["a2"], p["net.7.weight"], dlog)
    do2 = da2 * (c["o2"] > 0)
    dh2, g["net.5.weight"], g["net.5.bias"] = bn_bwd(do2, p["net.5.weight"], c["bc2"], (0,))
    da1d, g["net.4.weight"], g["net.4.bias"] = linear_bwd(c["a1d"], p["net.4.weight"], dh2)
    if c["train"] and c["drop_mask"] is not None:
        da1 = da1d * c["drop_mask"] / np.float32(1.0 - c["p_drop"])
    else:
        da1 = da1d
    do1 = da1 * (c["o1"] > 0)
    dh1, g["net.1.weight"], g["net.1.bias"] = bn_bwd(do1, p["net.1.weight"], c["bc1"], (0,))
    dx, g["net.0.weight"], g["net.0.bias"] = linear_bwd(c["x"], p["net.0.weight"], dh1)
    g["dx"] = dx
    return loss, g


def mlp_train_step(p, state, x, labels, lr, drop_mask=None, weight_decay=1e-4):
    """One iteration of the MLP batch loop (R.md:2639-2646)."""
    out = mlp_forward(p, x, True, drop_mask=drop_mask)
    loss, g = mlp_backward(p, out, labels)
    for k, v in out["new_buffers"].items():
        p[k] = v
    for k in list(p.keys()):
        if k.endswith("num_batches_tracked"):
            p[k] = p[k] + 1
    adam_step(p, g, state, lr, weight_decay=weight_decay)
    return loss, out, g


def new_adam_state():
    return {"step": 0, "m": {}, "v": {}}
