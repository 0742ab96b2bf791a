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
    nt = lib.eae_op_conv_s2_ntiles(0, B, H, H)
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
    nt = lib.eae_op_conv_s2_ntiles(1, B, H, H)
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
    wp = np.zeros((32, 32), np.float32)
    wp[:, :27] = wq.reshape(32, 3, 9).transpose(0, 2, 1).reshape(32, 27)       # k = tap*3 + c
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
