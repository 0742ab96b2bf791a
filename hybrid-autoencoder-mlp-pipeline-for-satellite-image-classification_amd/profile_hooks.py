"""Per-kernel timing for bench.py's `roofline` object: HIP events recorded by the engine on the launch stream around the
dominant kernel (enc.conv2 forward = conv_s2_kernel<32,64,...>) inside real train steps."""
from __future__ import annotations

import ctypes as C

import torch

from ._lib import check

# enc.conv2 forward, per image (DESIGN.md "Roofline accounting"): reads the 32x32x32 bf16 input once, writes the
# 16x16x64 bf16 output once; 256 output pixels x K=288 x N=64 MACs.
CONV2_BYTES_PER_IMG = 32 * 32 * 32 * 2 + 16 * 16 * 64 * 2
CONV2_FLOP_PER_IMG = 2.0 * 256 * 288 * 64


def dominant_kernel_roofline(eng, step_fn, batch, hbm_peak_gbs, mfma_peak_tflops, steps=32):
    check(eng.lib.eae_profile_enable(eng.ctx, 1))
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    tot, emp, n = C.c_double(), C.c_double(), C.c_longlong()
    check(eng.lib.eae_profile_read2(eng.ctx, C.byref(tot), C.byref(emp), C.byref(n)))
    check(eng.lib.eae_profile_enable(eng.ctx, 0))
    bracket_us = 1e3 * tot.value / max(n.value, 1)
    empty_us = 1e3 * emp.value / max(n.value, 1)
    # a HIP event bracket around ONE launch also times the two event records (an empty bracket recorded right after each
    # timed one measures them); the kernel's own duration -- what rocprofv3 reports -- is the difference
    us = max(bracket_us - empty_us, 1e-3)
    byts = batch * CONV2_BYTES_PER_IMG
    achieved = byts / (us * 1e-6) / 1e9
    return {"kernel": "conv_s2_kernel<32,64,...> (enc.conv2 forward: implicit GEMM M=B*256, K=288, N=64)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": hbm_peak_gbs, "unit": "GB/s",
            "frac": round(achieved / hbm_peak_gbs, 4), "traffic": None,
            "avg_launch_us": round(us, 2), "event_bracket_us": round(bracket_us, 2), "empty_bracket_us": round(empty_us, 2),
            "launches_timed": int(n.value),
            "algorithmic_bytes_per_launch": byts,
            "tflops": round(batch * CONV2_FLOP_PER_IMG / (us * 1e-6) / 1e12, 1)}
