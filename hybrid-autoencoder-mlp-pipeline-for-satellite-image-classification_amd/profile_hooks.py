"""Per-kernel timing hooks for bench.py's `roofline` object (HIP events on the launch stream)."""
from __future__ import annotations

import torch


def dominant_kernel_roofline(eng, x, y, reps, hbm_peak_gbs, mfma_peak_tflops):
    raise NotImplementedError("filled in once the rocprof summary names the dominant kernel")
