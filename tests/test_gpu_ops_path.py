"""Op-level parity of EVERY conv-family instantiation on the training path, through the C ABI's per-op entry points.

The end-to-end tests reach these kernels only behind 20 other layers of bf16 noise; here each one runs alone on exact
bf16 operands against the NumPy oracle (R.md:292-305 / 370-379 and their autograd), with
  * the load transforms the path uses (mode 1: BatchNorm-apply + ReLU on load, mode 2: BatchNorm-backward-apply on load),
  * the epilogues the path uses (0: bias + batch statistics, 1: ReLU mask + BatchNorm-backward sums, 2: plain),
  * the map sizes the path uses (32/16/8/4 pixels: 16x8 tiles, 2 images per tile, 8 images per tile) with batch sizes that
    are NOT a multiple of the images-per-tile count.
"""
import os
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
    from eae_amd._lib import check
    a, b = w.shape[0], w.shape[1]
    wd = G.f32(w)
    buf = torch.empty(2 * a * b * 9, dtype=torch.bfloat16, device=G.dev())
    p1, p2 = buf[: a * b * 9], buf[a * b * 9:]
    check(lib.eae_op_pack3x3(G.stream(), G.ptr(wd), a, b, G.ptr(p1), G.ptr(p2)))
    torch.cuda.synchronize()
    return buf, p1, p2


def _f32(a):
    return np.asarray(a, np.float64).astype(np.float32)


class Src:
    """A logical activation tensor the way a kernel materialises it on load (eae_src), plus its NumPy value."""

    def __init__(self, mode, shape, rng, scale=1.0):
        import gpu_util as G
        c = shape[1]
        self.mode = mode
        self.keep = []
        bc = lambda v: v[None, :, None, None].astype(np.float64)
        if mode in (0, 4):  # as stored (4 = a stored gradient tensor: the same load, e5m2 in the fp8 variants)
            x = O.bf16_round((rng.standard_normal(shape) * scale).astype(np.float32))
            d = G.to_nhwc_bf16(x)
            self.keep = [d]
            self.src = G.src(mode, d)
            self.value = x
        elif mode == 1:     # max(0, s*y + t), fp32 fma, rounded to bf16
            y = O.bf16_round((rng.standard_normal(shape) * scale).astype(np.float32))
            s = (1.0 + 0.2 * rng.standard_normal(c)).astype(np.float32)
            t = (0.3 * rng.standard_normal(c)).astype(np.float32)
            coef = np.stack([s, t, np.zeros(c, np.float32), np.ones(c, np.float32)])
            d, cd = G.to_nhwc_bf16(y), G.f32(coef)
            self.keep = [d, cd]
            self.src = G.src(1, d, None, cd)
            self.value = O.bf16_round(np.maximum(_f32(y.astype(np.float64) * bc(s) + bc(t)), 0.0))
        else:               # A*g + (B*y + C), two fp32 fmas, rounded to bf16
            g = O.bf16_round((rng.standard_normal(shape) * scale).astype(np.float32))
            y = O.bf16_round(rng.standard_normal(shape).astype(np.float32))
            a_ = (1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)
            b_ = (0.1 * rng.standard_normal(c)).astype(np.float32)
            c_ = (0.05 * rng.standard_normal(c)).astype(np.float32)
            coef = np.stack([a_, b_, c_])
            gd, yd, cd = G.to_nhwc_bf16(g), G.to_nhwc_bf16(y), G.f32(coef)
            self.keep = [gd, yd, cd]
            self.src = G.src(2, gd, yd, cd)
            inner = _f32(y.astype(np.float64) * bc(b_) + bc(c_))
            self.value = O.bf16_round(_f32(g.astype(np.float64) * bc(a_) + inner.astype(np.float64)))


# (kind, cin, cout, input size, source mode, epilogue, batch): every igemm_s2_kernel instantiation of eae_conv_launch.hip at
# the map size it runs on for 64x64 images
IGEMM_CASES = [
    (0, 32, 64, 32, 1, 0, 3), (0, 64, 128, 16, 1, 0, 3), (0, 128, 256, 8, 1, 0, 11),          # enc.conv2/3/4 forward
    (0, 32, 64, 32, 2, 1, 2), (0, 64, 128, 16, 2, 1, 5), (0, 128, 256, 8, 2, 2, 13),          # backward-data of dec.deconv3/2/1
    (1, 256, 128, 4, 0, 0, 7), (1, 128, 64, 8, 1, 0, 3), (1, 64, 32, 16, 1, 0, 2),            # dec.deconv1/2/3 forward
    (1, 256, 128, 4, 2, 1, 5), (1, 128, 64, 8, 2, 1, 3), (1, 64, 32, 16, 2, 1, 3),            # backward-data of enc.conv4/3/2
]


@pytest.mark.parametrize("kind,cin,cout,hin,smode,epi,B", IGEMM_CASES)
def test_igemm_instantiation(lib, kind, cin, cout, hin, smode, epi, B):
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(1000 + 37 * cin + 11 * kind + smode + 3 * epi)
    src = Src(smode, (B, cin, hin, hin), rng)
    hout = hin // 2 if kind == 0 else hin * 2
    if kind == 0:
        w = O.bf16_round((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32))
        buf, p1, p2 = _pack(lib, w)
        wp = p1                                     # [cout][9][cin]
        ref = O.conv_s2_fwd(src.value, w, None)
    else:
        w = O.bf16_round((rng.standard_normal((cin, cout, 3, 3)) / np.sqrt(9 * cin / 4)).astype(np.float32))
        buf, p1, p2 = _pack(lib, w)
        wp = p2                                     # [cout][9][cin]
        ref = O.deconv_s2_fwd(src.value, w, None)
    bias = rng.standard_normal(cout).astype(np.float32) if epi == 0 else None
    bd = G.f32(bias) if bias is not None else None
    out = torch.zeros((B, hout, hout, cout), dtype=torch.bfloat16, device=G.dev())
    nt = lib.eae_op_conv_s2_ntiles(kind, cin, B, hin, hin)
    part = torch.zeros((2, cout, nt), dtype=torch.float32, device=G.dev())
    yprev_d = pcoef_d = None
    if epi == 1:
        yprev = O.bf16_round(rng.standard_normal((B, cout, hout, hout)).astype(np.float32))
        ps = (1.0 + 0.2 * rng.standard_normal(cout)).astype(np.float32)
        pt = (0.3 * rng.standard_normal(cout)).astype(np.float32)
        pm = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        pi = (1.0 + 0.2 * rng.random(cout)).astype(np.float32)
        yprev_d, pcoef_d = G.to_nhwc_bf16(yprev), G.f32(np.stack([ps, pt, pm, pi]))
    check(lib.eae_op_conv_s2(G.stream(), kind, src.src, cin, cout, B, hin, hin, G.ptr(wp), G.ptr(bd), G.ptr(out),
                             G.ptr(part) if epi != 2 else None, epi, G.ptr(yprev_d), G.ptr(pcoef_d)))
    torch.cuda.synchronize()
    got = G.from_nhwc(out)
    bc = lambda v: v[None, :, None, None]
    if epi == 0:
        ref = ref + bc(bias)
    tol = 2 ** -7 * np.abs(ref).max() + 1e-3
    if epi == 1:
        act = _f32(yprev.astype(np.float64) * bc(ps) + bc(pt))
        mask = act > 0
        # a flipped mask bit needs |act| at fp32 rounding level: compare where the decision is clear
        clear = np.abs(act) > 1e-5
        assert np.abs(got - ref * mask)[clear].max() <= tol, G.relmax(got, ref * mask)
        assert np.all(got[clear & ~mask] == 0.0)
        xhat = _f32(yprev.astype(np.float64) * bc(pi) + bc(-pm * pi))
        s = part.cpu().numpy().sum(2)
        np.testing.assert_allclose(s[0], got.astype(np.float64).sum((0, 2, 3)), rtol=2e-4, atol=2e-2)
        np.testing.assert_allclose(s[1], (got.astype(np.float64) * xhat).sum((0, 2, 3)), rtol=2e-4, atol=2e-2)
    else:
        assert np.abs(got - ref).max() <= tol, G.relmax(got, ref)
        if epi == 0:
            s = part.cpu().numpy().sum(2)
            np.testing.assert_allclose(s[0], got.astype(np.float64).sum((0, 2, 3)), rtol=2e-4, atol=2e-2)
            np.testing.assert_allclose(s[1], (got.astype(np.float64) ** 2).sum((0, 2, 3)), rtol=2e-4, atol=2e-2)


# (cs, cb, small-map size, small mode, big mode, batch): every wgrad_s2_kernel instantiation of eae_wgrad_launch.hip at the
# map size it runs on for 64x64 images, plus batch sizes that leave the last tile / the last workgroup partly empty
WGRAD_CASES = [
    (64, 32, 16, 2, 1, 3), (128, 64, 8, 2, 1, 5), (256, 128, 4, 2, 1, 11),       # enc.conv2/3/4
    (64, 32, 16, 1, 2, 2), (128, 64, 8, 1, 2, 3), (256, 128, 4, 0, 2, 13),       # dec.deconv3/2/1
    (64, 32, 16, 0, 0, 1), (128, 64, 8, 2, 1, 1), (256, 128, 4, 2, 1, 3),        # a single tile, fewer images than a tile holds
    (64, 32, 16, 2, 1, 37), (256, 128, 4, 0, 2, 70),                             # several tiles per workgroup (odd counts)
    # the train step's forms: the gradient operand is the stored dy tensor (mode 4) the backward-data kernel wrote
    (64, 32, 16, 4, 1, 3), (128, 64, 8, 4, 1, 5), (256, 128, 4, 4, 1, 11),       # enc.conv2/3/4
    (64, 32, 16, 1, 4, 2), (128, 64, 8, 1, 4, 3), (256, 128, 4, 0, 4, 13),       # dec.deconv3/2/1
    (128, 64, 8, 4, 1, 1), (256, 128, 4, 4, 1, 3), (64, 32, 16, 4, 1, 37), (256, 128, 4, 0, 4, 70), (128, 64, 8, 1, 4, 67),
    (64, 32, 8, 4, 1, 5), (64, 32, 4, 1, 4, 9), (128, 64, 16, 4, 1, 2), (256, 128, 16, 0, 4, 1), (128, 64, 4, 1, 4, 17),   # the other tile geometries
]


@pytest.mark.parametrize("cs,cb,hs,smode,bmode,B", WGRAD_CASES)
def test_wgrad_instantiation(lib, cs, cb, hs, smode, bmode, B):
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(2000 + cs + 7 * smode + 3 * bmode + B)
    small = Src(smode, (B, cs, hs, hs), rng)
    big = Src(bmode, (B, cb, 2 * hs, 2 * hs), rng)
    scratch = torch.empty(6 * 1024 * 1024, dtype=torch.float32, device=G.dev())
    dw = torch.full((cs, cb, 3, 3), float("nan"), dtype=torch.float32, device=G.dev())
    check(lib.eae_op_wgrad_s2(G.stream(), small.src, big.src, cs, cb, B, hs, hs, G.ptr(scratch), scratch.numel(), G.ptr(dw)))
    torch.cuda.synchronize()
    _, ref, _ = O.conv_s2_bwd(big.value, np.zeros((cs, cb, 3, 3), np.float32), small.value)
    got = dw.cpu().numpy()
    assert np.isfinite(got).all()
    # exact bf16 operands, fp32 accumulation: only the summation order differs (a transformed operand can differ from the
    # NumPy value by one bf16 ulp where the fp32 fma lands on a rounding tie)
    assert G.relmax(got, ref) < (1e-4 if smode in (0, 4) and bmode in (0, 4) else 1.5e-3), G.relmax(got, ref)


def test_wgrad_repeatable_and_slice_count_independent(lib):
    """Deterministic two-stage reduction: the same call twice is bitwise equal, and the result does not depend on how many
    workgroups (position slices) the launcher picks beyond fp32 summation order."""
    import gpu_util as G
    from eae_amd._lib import check
    rng = np.random.default_rng(5)
    B, cs, cb, hs = 9, 128, 64, 8
    small = Src(2, (B, cs, hs, hs), rng)
    big = Src(1, (B, cb, 2 * hs, 2 * hs), rng)
    scratch = torch.empty(6 * 1024 * 1024, dtype=torch.float32, device=G.dev())
    outs = []
    for _ in range(2):
        dw = torch.zeros((cs, cb, 3, 3), dtype=torch.float32, device=G.dev())
        scratch.fill_(float("nan"))
        check(lib.eae_op_wgrad_s2(G.stream(), small.src, big.src, cs, cb, B, hs, hs, G.ptr(scratch), scratch.numel(), G.ptr(dw)))
        torch.cuda.synchronize()
        outs.append(dw.cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])


def test_igemm_cases_on_the_large_tile_geometries_and_the_wave_specialised_kernel():
    """The launcher picks the geometry from the grid size (small batches, like the cases above: 64-position tiles / 32-channel
    blocks) and reads EAE_IGEMM2 / EAE_IG_SMALL once per process.  The cases above therefore run again in ONE child process per
    setting: large tiles (EAE_IG_SMALL=0, what B=512 uses) with the wave-specialised kernel on every layer it is instantiated
    for (EAE_IGEMM2=2) and on none (0)."""
    import subprocess
    import sys
    for mode in ("2", "0"):
        env = dict(os.environ, EAE_IGEMM2=mode, EAE_IG_SMALL="0")
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "igemm_instantiation",
                            "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"EAE_IGEMM2={mode}\n" + r.stdout[-3000:] + r.stderr[-2000:]
