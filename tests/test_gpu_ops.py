"""Kernel-level parity (through the C ABI's per-op entry points) against the NumPy oracle.

bf16 operands / fp32 accumulation: the oracle is fed the same bf16-rounded operands, so the only differences are
accumulation order and the final bf16 rounding of the stored output (<= 1 bf16 ulp = 2^-8 relative)."""
import numpy as np
import pytest
import torch

from oracle import ae_numpy as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from eae_amd import _lib
    return _lib.load()


def _pack(lib, w):
    import gpu_util as G
    a, b = w.shape[0], w.shape[1]
    wd = G.f32(w)
    buf = torch.empty(2 * a * b * 9, dtype=torch.bfloat16, device=G.dev())
    p1, p2 = buf[: a * b * 9], buf[a * b * 9:]
    from eae_amd._lib import check
    check(lib.eae_op_pack3x3(G.stream(), G.ptr(wd), a, b, G.ptr(p1), G.ptr(p2)))
    torch.cuda.synchronize()
    return buf, p1, p2


def test_pack3x3(lib):
    rng = np.random.default_rng(0)
    w = rng.standard_normal((64, 32, 3, 3)).astype(np.float32)
    buf, p1, p2 = _pack(lib, w)
    wq = O.bf16_round(w)
    e1 = wq.reshape(64, 32, 9).transpose(0, 2, 1)        # [A][9][B]
    e2 = wq.reshape(64, 32, 9).transpose(1, 2, 0)        # [B][9][A]
    assert np.array_equal(p1.float().cpu().numpy().reshape(64, 9, 32), e1)
    assert np.array_equal(p2.float().cpu().numpy().reshape(32, 9, 64), e2)


@pytest.mark.parametrize("B,H", [(3, 32), (5, 64)])
def test_conv_s2_raw_fwd(lib, B, H):
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(1)
    x = O.bf16_round(rng.standard_normal((B, 32, H, H)).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((64, 32, 3, 3)) * 0.1).astype(np.float32))
    b = rng.standard_normal(64).astype(np.float32)
    buf, p1, p2 = _pack(lib, w)
    xd = G.to_nhwc_bf16(x)
    out = torch.empty((B, H // 2, H // 2, 64), dtype=torch.bfloat16, device=G.dev())
    nt = lib.eae_op_conv_s2_ntiles(0, 32, B, H, H)
    part = torch.zeros((2, 64, nt), dtype=torch.float32, device=G.dev())
    bd = G.f32(b)
    check(lib.eae_op_conv_s2(G.stream(), 0, G.src(0, xd), 32, 64, B, H, H, G.ptr(p1), G.ptr(bd), G.ptr(out), G.ptr(part), 0, None, None))
    torch.cuda.synchronize()
    ref = O.conv_s2_fwd(x, w, b)
    got = G.from_nhwc(out)
    assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max() + 1e-3, G.relmax(got, ref)
    ps = part.cpu().numpy().sum(2)
    np.testing.assert_allclose(ps[0], got.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(ps[1], (got.astype(np.float64) ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("B,H", [(3, 16), (2, 32)])
def test_deconv_s2_raw_fwd(lib, B, H):
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(2)
    x = O.bf16_round(rng.standard_normal((B, 64, H, H)).astype(np.float32))
    w = O.bf16_round((rng.standard_normal((64, 32, 3, 3)) * 0.1).astype(np.float32))      # [Cin][Cout][3][3]
    b = rng.standard_normal(32).astype(np.float32)
    buf, p1, p2 = _pack(lib, w)
    xd = G.to_nhwc_bf16(x)
    out = torch.empty((B, 2 * H, 2 * H, 32), dtype=torch.bfloat16, device=G.dev())
    nt = lib.eae_op_conv_s2_ntiles(1, 64, B, H, H)
    part = torch.zeros((2, 32, nt), dtype=torch.float32, device=G.dev())
    bd = G.f32(b)
    check(lib.eae_op_conv_s2(G.stream(), 1, G.src(0, xd), 64, 32, B, H, H, G.ptr(p2), G.ptr(bd), G.ptr(out), G.ptr(part), 0, None, None))
    torch.cuda.synchronize()
    ref = O.deconv_s2_fwd(x, w, b)
    got = G.from_nhwc(out)
    assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max() + 1e-3, G.relmax(got, ref)
    ps = part.cpu().numpy().sum(2)
    np.testing.assert_allclose(ps[0], got.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


def test_wgrad_s2_raw(lib):
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(3)
    B, Hs = 3, 16
    big = O.bf16_round(rng.standard_normal((B, 32, 2 * Hs, 2 * Hs)).astype(np.float32))
    small = O.bf16_round(rng.standard_normal((B, 64, Hs, Hs)).astype(np.float32))
    bd_, sd_ = G.to_nhwc_bf16(big), G.to_nhwc_bf16(small)
    scratch = torch.empty(4 * 1024 * 1024, dtype=torch.float32, device=G.dev())
    dw = torch.zeros((64, 32, 3, 3), dtype=torch.float32, device=G.dev())
    check(lib.eae_op_wgrad_s2(G.stream(), G.src(0, sd_), G.src(0, bd_), 64, 32, B, Hs, Hs, G.ptr(scratch), scratch.numel(), G.ptr(dw)))
    torch.cuda.synchronize()
    _, ref, _ = O.conv_s2_bwd(big, np.zeros((64, 32, 3, 3), np.float32), small)
    assert G.relmax(dw.cpu().numpy(), ref) < 1e-4


def test_edge_conv_and_wgrad(lib):
    import gpu_util as G
    from eae_amd._lib import check
    from eae_amd import _lib as L
    rng = np.random.default_rng(4)
    B = 3
    x = rng.random((B, 3, 64, 64)).astype(np.float32)
    w = (rng.standard_normal((32, 3, 3, 3)) * 0.2).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    wq = O.bf16_round(w)
    wp = np.zeros((32, 64), np.float32)
    wp4 = np.zeros((32, 9, 4), np.float32)
    wp4[:, :, :3] = wq.reshape(32, 3, 9).transpose(0, 2, 1)                    # k = tap*4 + c (4th channel and k >= 36: zero)
    wp[:, :36] = wp4.reshape(32, 36)
    wpd = G.f32(wp).to(torch.bfloat16)
    xd, bd = G.f32(x), G.f32(b)
    out = torch.empty((B, 32, 32, 32), dtype=torch.bfloat16, device=G.dev())
    nt = B * 8
    part = torch.zeros((2, 32, nt), dtype=torch.float32, device=G.dev())
    check(lib.eae_op_edge_conv(G.stream(), 0, G.ptr(xd), B, 64, 64, G.ptr(wpd), G.ptr(bd), G.ptr(out), G.ptr(part), 0, None, None))
    torch.cuda.synchronize()
    ref = O.conv_s2_fwd(O.bf16_round(x), wq, b)
    got = G.from_nhwc(out)
    assert np.abs(got - ref).max() <= 2 ** -7 * np.abs(ref).max() + 1e-3, G.relmax(got, ref)
    np.testing.assert_allclose(part.cpu().numpy().sum(2)[0], got.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    # weight gradient with a plain (raw) side tensor
    dy = O.bf16_round(rng.standard_normal((B, 32, 32, 32)).astype(np.float32))
    dyd = G.to_nhwc_bf16(dy)
    scratch = torch.empty(1024 * 1024, dtype=torch.float32, device=G.dev())
    dw = torch.zeros((32, 3, 3, 3), dtype=torch.float32, device=G.dev())
    check(lib.eae_op_edge_wgrad(G.stream(), 0, G.ptr(xd), B, 64, 64, G.src(0, dyd), G.ptr(scratch), scratch.numel(), G.ptr(dw)))
    torch.cuda.synchronize()
    _, refw, _ = O.conv_s2_bwd(O.bf16_round(x), wq, dy)
    assert G.relmax(dw.cpu().numpy(), refw) < 1e-4


def test_adam_kernel(lib):
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(5)
    n = 4096
    p = {"w": rng.standard_normal(n).astype(np.float32)}
    st = O.new_adam_state()
    pd = G.f32(p["w"])
    md, vd = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        O.adam_step(p, {"w": g}, st, 1e-3, weight_decay=1e-4)
        check(lib.eae_op_adam(G.stream(), G.ptr(pd), G.ptr(G.f32(g)), G.ptr(md), G.ptr(vd), n, 1e-3, 0.9, 0.999, 1e-8, 1e-4, step))
    torch.cuda.synchronize()
    np.testing.assert_allclose(pd.cpu().numpy(), p["w"], rtol=1e-5, atol=1e-6)


def test_augment_matches_oracle(lib):
    """eae_augment (flip -> pad-4 crop -> /255 -> + 0.03*noise) vs the NumPy restatement with the same explicit draws."""
    import gpu_util as G
    from eae_amd.augment import augment_batch
    from oracle.augment_numpy import augment_ref
    rng = np.random.default_rng(11)
    b = 7
    u8 = rng.integers(0, 256, (b, 64, 64, 3), dtype=np.uint8)
    flips = rng.integers(0, 2, b); tops = rng.integers(0, 9, b); lefts = rng.integers(0, 9, b)
    noise = rng.standard_normal((b, 3, 64, 64)).astype(np.float32)
    params = torch.from_numpy(np.stack([flips, tops, lefts], 1).astype(np.int32))
    got = augment_batch(torch.from_numpy(u8).cuda(), train=True, params=params, noise=torch.from_numpy(noise)).cpu().numpy()
    ref = augment_ref(u8, flips, tops, lefts, noise)
    assert np.abs(got - ref).max() < 1e-6
    got = augment_batch(torch.from_numpy(u8).cuda(), train=False).cpu().numpy()
    assert np.array_equal(got, u8.transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255.0))
    # device RNG mode: right distributions, reproducible for a fixed (seed, step), different across steps
    big = torch.from_numpy(rng.integers(0, 256, (256, 64, 64, 3), dtype=np.uint8)).cuda()
    a = augment_batch(big, train=True, seed=3, step=5)
    b2 = augment_batch(big, train=True, seed=3, step=5)
    c = augment_batch(big, train=True, seed=3, step=6)
    assert torch.equal(a, b2) and not torch.equal(a, c)
    clean = augment_batch(big, train=True, seed=3, step=5, noise_std=0.0)
    resid = (a - clean).cpu().numpy()
    assert abs(resid.std() - 0.03) < 5e-4 and abs(resid.mean()) < 2e-4


def _nhwc_perm_rows(w_ref_KL, Pn):
    """reference dec.fc weight [K][L] with K = c*Pn + p  ->  packed rows k' = p*256 + c"""
    K, L = w_ref_KL.shape
    return w_ref_KL.reshape(256, Pn, L).transpose(1, 0, 2).reshape(K, L)


def test_fc_ops(lib):
    """enc.fc forward (split-K, BN+ReLU source), dec.fc forward (bias, bf16 out) and both weight gradients (R.md:309, 365)."""
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(11)
    B, L, Pn = 37, 64, 16
    K = 256 * Pn
    # ---- enc.fc forward: z = relu(s*y4+t) (NHWC-flattened) . We^T + b
    y4 = O.bf16_round(rng.standard_normal((B, 256, 4, 4)).astype(np.float32))
    s = (1.0 + 0.1 * rng.standard_normal(256)).astype(np.float32); t = (0.1 * rng.standard_normal(256)).astype(np.float32)
    coef = np.stack([s, t, np.zeros(256, np.float32), np.ones(256, np.float32)])
    act = O.bf16_round(np.maximum(y4 * s[None, :, None, None] + t[None, :, None, None], 0.0))        # what the kernel feeds the MFMA
    we = O.bf16_round((rng.standard_normal((L, K)) * 0.02).astype(np.float32))                        # reference layout [L][c*Pn+p]
    be = rng.standard_normal(L).astype(np.float32)
    we_pack = we.reshape(L, 256, Pn).transpose(0, 2, 1).reshape(L, K)                                 # k' = p*256 + c
    y4d = G.to_nhwc_bf16(y4); coefd = G.f32(coef); wed = G.f32(we_pack).to(torch.bfloat16); bed = G.f32(be)
    scratch = torch.empty((K // 128) * B * L + 1024, dtype=torch.float32, device=G.dev())
    z = torch.empty((B, L), dtype=torch.float32, device=G.dev())
    check(lib.eae_op_fc_splitk(G.stream(), G.src(1, y4d, None, coefd), G.ptr(wed), B, L, K, G.ptr(bed), None, G.ptr(scratch), scratch.numel(), G.ptr(z)))
    torch.cuda.synchronize()
    ref = O.linear_fwd(act.reshape(B, K).astype(np.float64), we.astype(np.float64), be)
    assert G.relmax(z.cpu().numpy(), ref) < 2e-3        # the kernel's BN+ReLU is fp32 then bf16: up to 1 ulp per operand vs the rounded oracle input
    # ---- dec.fc forward: d0 (NHWC bf16) = z . Wd^T + bd
    zz = rng.standard_normal((B, L)).astype(np.float32)
    wd = O.bf16_round((rng.standard_normal((K, L)) * 0.1).astype(np.float32))                         # reference [K][L], K = c*Pn+p
    bd = rng.standard_normal(K).astype(np.float32)
    wdd = G.f32(_nhwc_perm_rows(wd, Pn)).to(torch.bfloat16)
    bdd = G.f32(bd.reshape(256, Pn).T.reshape(K))
    zd = G.f32(zz)
    d0 = torch.empty((B, K), dtype=torch.bfloat16, device=G.dev())
    check(lib.eae_op_fc_bias_bf16(G.stream(), G.ptr(zd), G.ptr(wdd), B, K, L, G.ptr(bdd), G.ptr(d0)))
    torch.cuda.synchronize()
    ref_d0 = O.linear_fwd(O.bf16_round(zz).astype(np.float64), wd.astype(np.float64), bd)           # [B][c*Pn+p]
    got_d0 = d0.float().cpu().numpy().reshape(B, Pn, 256).transpose(0, 2, 1).reshape(B, K)
    assert np.abs(got_d0 - ref_d0).max() <= 2 ** -7 * np.abs(ref_d0).max() + 1e-3
    # ---- dec.fc weight gradient (mode 0): dW[K][L] = g^T . z, db = sum g
    g = O.bf16_round(rng.standard_normal((B, 256, 4, 4)).astype(np.float32))
    gd = G.to_nhwc_bf16(g)
    dw = torch.zeros((K, L), dtype=torch.float32, device=G.dev()); db = torch.zeros(K, dtype=torch.float32, device=G.dev())
    check(lib.eae_op_fc_wgrad(G.stream(), 0, G.src(0, gd), G.src(3, zd), B, K, L, Pn, G.ptr(dw), G.ptr(db)))
    torch.cuda.synchronize()
    _, refw, refb = O.linear_bwd(O.bf16_round(zz).astype(np.float64), wd.astype(np.float64), g.reshape(B, K).astype(np.float64))
    assert G.relmax(dw.cpu().numpy(), refw) < 1e-4 and G.relmax(db.cpu().numpy(), refb) < 1e-4
    # ---- enc.fc weight gradient (mode 1): dW[L][K] = dz^T . act, db = sum dz
    dz = rng.standard_normal((B, L)).astype(np.float32)
    dzd = G.f32(dz)
    dwe = torch.zeros((L, K), dtype=torch.float32, device=G.dev()); dbe = torch.zeros(L, dtype=torch.float32, device=G.dev())
    check(lib.eae_op_fc_wgrad(G.stream(), 1, G.src(3, dzd), G.src(1, y4d, None, coefd), B, L, K, Pn, G.ptr(dwe), G.ptr(dbe)))
    torch.cuda.synchronize()
    _, refwe, refbe = O.linear_bwd(act.reshape(B, K).astype(np.float64), we.astype(np.float64), O.bf16_round(dz).astype(np.float64))
    assert G.relmax(dwe.cpu().numpy(), refwe) < 3e-3 and G.relmax(dbe.cpu().numpy(), refbe) < 1e-2


def test_head_ce_op(lib):
    """Linear(L,128)-ReLU-Linear(128,C) + CrossEntropy(mean) forward/backward in fp32 (R.md:423-427, 623) vs the oracle."""
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(12)
    for B, L, C in ((21, 64, 10), (64, 128, 10)):
        z = rng.standard_normal((B, L)).astype(np.float32)
        w1 = (rng.standard_normal((128, L)) * 0.2).astype(np.float32); b1 = (rng.standard_normal(128) * 0.1).astype(np.float32)
        w2 = (rng.standard_normal((C, 128)) * 0.2).astype(np.float32); b2 = (rng.standard_normal(C) * 0.1).astype(np.float32)
        labels = rng.integers(0, C, B).astype(np.int64)
        d = [G.f32(a) for a in (z, w1, b1, w2, b2)]
        lab = torch.from_numpy(labels).to(G.dev())
        nsc = lib.eae_op_head_scratch_floats(B, L, C)
        scratch = torch.zeros(nsc, dtype=torch.float32, device=G.dev())
        logits = torch.empty((B, C), dtype=torch.float32, device=G.dev()); dz = torch.empty((B, L), dtype=torch.float32, device=G.dev())
        r4 = lambda n: (n + 3) // 4 * 4
        grads = torch.zeros(r4(128 * L) + 128 + r4(128 * C) + r4(C), dtype=torch.float32, device=G.dev())
        loss2 = torch.zeros(2, dtype=torch.float32, device=G.dev())
        check(lib.eae_op_head_ce(G.stream(), *[G.ptr(t) for t in d], G.ptr(lab), B, L, C, G.ptr(logits), G.ptr(dz), G.ptr(grads), G.ptr(loss2),
                                 G.ptr(scratch), nsc))
        torch.cuda.synchronize()
        h = np.maximum(O.linear_fwd(z.astype(np.float64), w1.astype(np.float64), b1), 0.0)
        lg = O.linear_fwd(h, w2.astype(np.float64), b2)
        loss, dl = O.cross_entropy(lg, labels)
        dh, dw2, db2 = O.linear_bwd(h, w2.astype(np.float64), dl.astype(np.float64))
        dh = dh * (h > 0)
        dzr, dw1, db1 = O.linear_bwd(z.astype(np.float64), w1.astype(np.float64), dh)
        np.testing.assert_allclose(logits.cpu().numpy(), lg, rtol=1e-4, atol=1e-4)
        l2 = loss2.cpu().numpy()
        assert abs(l2[0] - loss) < 1e-4 and l2[1] == float((lg.argmax(1) == labels).sum())
        np.testing.assert_allclose(dz.cpu().numpy(), dzr, rtol=1e-3, atol=1e-6)
        gr = grads.cpu().numpy()
        o = 0
        for ref in (dw1, db1, dw2, db2):
            n = ref.size
            np.testing.assert_allclose(gr[o:o + n].reshape(ref.shape), ref, rtol=1e-3, atol=1e-6)
            o += r4(n)


def test_sigmoid_bwd_op(lib):
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(13)
    B = 2
    xh = rng.random((B, 3, 64, 64)).astype(np.float32) * 0.98 + 0.01
    dxh = rng.standard_normal((B, 3, 64, 64)).astype(np.float32)
    g4 = torch.zeros((B, 64, 64, 4), dtype=torch.bfloat16, device=G.dev())
    db = torch.zeros(4, dtype=torch.float32, device=G.dev())
    scratch = torch.zeros((B * 64 * 64 + 255) // 256 * 4 + 16, dtype=torch.float32, device=G.dev())
    xhd, dxhd = G.f32(xh), G.f32(dxh)
    check(lib.eae_op_sigmoid_bwd(G.stream(), G.ptr(xhd), G.ptr(dxhd), B, 64, 64, G.ptr(g4), G.ptr(db), G.ptr(scratch)))
    torch.cuda.synchronize()
    ref = dxh * xh * (1.0 - xh)
    got = g4.float().cpu().numpy()[..., :3].transpose(0, 3, 1, 2)
    assert np.abs(got - ref).max() <= 2 ** -8 * np.abs(ref).max() + 1e-6
    np.testing.assert_allclose(db.cpu().numpy()[:3], O.bf16_round(ref).sum((0, 2, 3)), rtol=2e-3, atol=1e-3)
