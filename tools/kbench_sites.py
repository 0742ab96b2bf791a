#!/usr/bin/env python3
"""Stand-alone duration of EVERY launch site (EAE_PROF_SITE(layer, role): forward / backward-data / weight gradient of the eight conv
layers) at a workload's shape, through the per-op C ABI with the source / epilogue modes the train step uses -- the kernel alone on
an otherwise idle GPU, launched back to back.  tools/kernel_report.py prints these beside the in-situ rocprofv3 averages so that
contention (side streams, the backward-data chain) can be told from kernel quality.

    python tools/kbench_sites.py <round> [b512 c2 c5bf16]     ->  profiles/<round>_standalone_<tag>.json  {site: us}

(bf16 kernels; the fp8 workload's table shows its bf16 twins' numbers.)  Diagnostic tool (GPU box)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G  # noqa: E402
from eae_amd import _lib  # noqa: E402
from eae_amd import profile_hooks as PH  # noqa: E402
from eae_amd._lib import check  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
WL = {"b512": (512, 64), "c2": (256, 64), "c5bf16": (128, 256)}
ENC_C = (3, 32, 64, 128, 256)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bf(shape):
    return (torch.randn(shape, device=dev) * 0.5).to(torch.bfloat16)


def coef(n, c):
    t = torch.randn((n, c), device=dev) * 0.1
    t[0] = 1.0 + t[0]
    return t.contiguous()


def igemm(kind, ci, co, B, hin, smode, epi):
    ho = hin // 2 if kind == 0 else hin * 2
    x = bf((B, hin, hin, ci))
    x2 = bf((B, hin, hin, ci)) if smode == 2 else None
    out = torch.empty((B, ho, ho, co), device=dev, dtype=torch.bfloat16)
    yprev = bf((B, ho, ho, co)) if epi == 1 else None
    pc = coef(4, co) if epi == 1 else None
    w = bf((co, 9, ci)); bias = torch.randn(co, device=dev)
    nt = lib.eae_op_conv_s2_ntiles(kind, ci, B, hin, hin)
    part = torch.zeros((nt, 2, co), device=dev)
    cf = coef(4 if smode == 1 else 3, ci)
    s = G.src(smode, x, x2, cf if smode else None)
    keep = (x, x2, out, yprev, pc, w, bias, part, cf)
    return timeit(lambda: check(lib.eae_op_conv_s2(G.stream(), kind, s, ci, co, B, hin, hin, G.ptr(w), G.ptr(bias), G.ptr(out),
                                                   G.ptr(part) if epi != 2 else None, epi, G.ptr(yprev), G.ptr(pc)))), keep


def wgrad(cs, cb, B, hs, smode, bmode, scratch):
    small = bf((B, hs, hs, cs)); small2 = bf((B, hs, hs, cs)) if smode == 2 else None
    big = bf((B, 2 * hs, 2 * hs, cb)); big2 = bf((B, 2 * hs, 2 * hs, cb)) if bmode == 2 else None
    dw = torch.zeros((cs, cb, 3, 3), device=dev)
    ss = G.src(smode, small, small2, coef(4 if smode == 1 else 3, cs) if smode else None)
    bs = G.src(bmode, big, big2, coef(4 if bmode == 1 else 3, cb) if bmode else None)
    return timeit(lambda: check(lib.eae_op_wgrad_s2(G.stream(), ss, bs, cs, cb, B, hs, hs, G.ptr(scratch), scratch.numel(), G.ptr(dw))))


def run(tag):
    B, hw = WL[tag]
    res = {}
    scratch = torch.empty(12 * 1024 * 1024, device=dev)
    # enc.conv2..4 (layer 1..3): forward, backward-data (transposed kernel on dy), weight gradient (+ its slice reduction)
    for layer in (1, 2, 3):
        ci, co, hin = ENC_C[layer], ENC_C[layer + 1], hw >> layer
        res[PH.prof_site(layer, 0)] = igemm(0, ci, co, B, hin, 1, 0)[0]
        res[PH.prof_site(layer, 1)] = igemm(1, co, ci, B, hin // 2, 2, 1)[0]
        res[PH.prof_site(layer, 2)] = wgrad(co, ci, B, hin // 2, 2, 1, scratch)
        torch.cuda.empty_cache()
    # dec.deconv1..3 (layer 4..6): forward (transposed kernel), backward-data (conv kernel on dy), weight gradient
    for i in (0, 1, 2):
        ci, co, hin = 256 >> i, 128 >> i, (hw >> 4) << i
        res[PH.prof_site(4 + i, 0)] = igemm(1, ci, co, B, hin, 0 if i == 0 else 1, 0)[0]
        res[PH.prof_site(4 + i, 1)] = igemm(0, co, ci, B, hin * 2, 2, 2 if i == 0 else 1)[0]
        res[PH.prof_site(4 + i, 2)] = wgrad(ci, co, B, hin, 0 if i == 0 else 1, 2, scratch)
        torch.cuda.empty_cache()
    # the edge layers
    x = torch.rand((B, 3, hw, hw), device=dev); wp = bf((32, 64)); bias = torch.randn(32, device=dev)
    out = bf((B, hw // 2, hw // 2, 32))
    ntile = B * (hw // 2 // 4) * (hw // 2 // 32)
    part = torch.zeros((ntile, 2, 32), device=dev)
    res[PH.prof_site(0, 0)] = timeit(lambda: check(lib.eae_op_edge_conv(G.stream(), 0, G.ptr(x), B, hw, hw, G.ptr(wp), G.ptr(bias), G.ptr(out), G.ptr(part), 0, None, None)))
    g1, y1 = bf((B, hw // 2, hw // 2, 32)), bf((B, hw // 2, hw // 2, 32))
    dw = torch.zeros((32, 3, 3, 3), device=dev)
    res[PH.prof_site(0, 2)] = timeit(lambda: check(lib.eae_op_edge_wgrad(G.stream(), 0, G.ptr(x), B, hw, hw, G.src(2, g1, y1, coef(3, 32)), G.ptr(scratch), scratch.numel(), G.ptr(dw))))
    a3 = bf((B, hw // 2, hw // 2, 32)); wj = bf((16, 128)); b3 = torch.randn(3, device=dev)
    g4 = torch.zeros((B, hw, hw, 4), device=dev, dtype=torch.bfloat16); lp = torch.zeros((ntile, 4), device=dev)
    cf = coef(4, 32)
    res[PH.prof_site(7, 0)] = timeit(lambda: check(lib.eae_op_deconv4_loss(G.stream(), G.src(1, a3, None, cf), B, hw // 2, hw // 2, G.ptr(wj), G.ptr(b3), G.ptr(x), 1e-3, None, G.ptr(g4), G.ptr(lp))))
    yprev = bf((B, hw // 2, hw // 2, 32)); pc = coef(4, 32)
    res[PH.prof_site(7, 1)] = timeit(lambda: check(lib.eae_op_edge_conv(G.stream(), 1, G.ptr(g4), B, hw, hw, G.ptr(wp), None, G.ptr(out), G.ptr(part), 1, G.ptr(yprev), G.ptr(pc))))
    res[PH.prof_site(7, 2)] = timeit(lambda: check(lib.eae_op_edge_wgrad(G.stream(), 1, G.ptr(g4), B, hw, hw, G.src(1, a3, None, cf), G.ptr(scratch), scratch.numel(), G.ptr(dw))))
    return res


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
    tags = sys.argv[2:] or list(WL)
    outdir = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}")
    os.makedirs(outdir, exist_ok=True)
    for tag in tags:
        res = run(tag)
        path = os.path.join(outdir, f"standalone_{tag}.json")
        json.dump({"workload": tag, "batch": WL[tag][0], "hw": WL[tag][1],
                   "note": "weight-gradient sites include their slice-reduction launch; back-to-back launches of the one kernel on an idle GPU",
                   "us": {str(k): round(v, 2) for k, v in sorted(res.items())}}, open(path, "w"), indent=1)
        print(tag, {PH.LAYER_NAMES[(k - 16) // 3] + " " + PH.ROLE_NAMES[(k - 16) % 3]: round(v, 1) for k, v in sorted(res.items())}, flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
