"""Per-kernel timing for bench.py's `roofline` object.

The engine can bracket ONE launch site with HIP events on the stream that launch goes to (eae_profile_enable(ctx, site),
include/eae.h).  `KERNELS` maps the exact rocprofv3 kernel name of every site to its algorithmic bytes / FLOPs per image
(SURVEY.md 8d minimal-traffic model: every logical tensor the kernel must read or write counted once, bf16 activations,
fp32 input image; the <= 0.6 MB of weights are L2-resident and excluded).  bench.py picks the site of the kernel with the
largest TotalDurationNs in the newest committed rocprofv3 summary under profiles/.
"""
from __future__ import annotations

import csv
import ctypes as C
import glob
import json
import os
import re
import subprocess

import torch

from ._lib import check

X, Y1, Y2 = 3 * 64 * 64, 32 * 32 * 32, 16 * 16 * 64          # elements per image: input, 32x32x32 map, 16x16x64 map
MF = 2.0 * 256 * 288 * 64 / 1e6                               # 9.44 MFLOP: one 3x3 stride-2 layer between 32 and 64 channels
EDGE_MF = 2.0 * 1024 * 27 * 32 / 1e6

# rocprofv3 kernel name -> (profile site, role, algorithmic bytes per image, MFLOP per image)
KERNELS = {
    "void igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 1, 0>(ConvArgs)": (1, "enc.conv2 forward", 2 * Y1 + 2 * Y2, MF),
    "void igemm_s2_kernel<1, 64, 32, 32, 16, 8, 1, 2, 1>(ConvArgs)": (2, "enc.conv2 backward-data (reads g, y of conv2's output and y1 for the ReLU mask, writes g1)", 2 * 2 * Y2 + 2 * Y1 + 2 * Y1, MF),
    "void igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 2, 1>(ConvArgs)": (3, "dec.deconv3 backward-data", 2 * 2 * Y1 + 2 * Y2 + 2 * Y2, MF),
    "void wgrad_s2_kernel<64, 32, 16, 8, 1, 2, 1>(WgradArgs)": (4, "enc.conv2 weight gradient (reads g, y of the 16x16x64 map and the 32x32x32 input map)", 2 * 2 * Y2 + 2 * Y1, MF),
    "void wgrad_s2_kernel<64, 32, 16, 8, 1, 1, 2>(WgradArgs)": (5, "dec.deconv3 weight gradient (reads the 16x16x64 input map and g, y of the 32x32x32 output map)", 2 * Y2 + 2 * 2 * Y1, MF),
    "void deconv4_loss_kernel<1>(Deconv4Args)": (6, "dec.deconv4 + sigmoid + MSE + its gradient", 2 * Y1 + 4 * X + 2 * 4 * 64 * 64, EDGE_MF),
    "void edge_wgrad_kernel<0, 2>(EdgeWgradArgs)": (7, "enc.conv1 weight gradient", 4 * X + 2 * 2 * Y1, EDGE_MF),
    "void edge_conv_kernel<1, 1>(EdgeArgs)": (8, "dec.deconv4 backward-data (+ReLU mask, BN-backward sums)", 2 * 4 * 64 * 64 + 2 * Y1 + 2 * Y1, EDGE_MF),
    "void igemm_s2_kernel<1, 64, 32, 32, 16, 8, 1, 1, 0>(ConvArgs)": (9, "dec.deconv3 forward", 2 * Y2 + 2 * Y1, MF),
}

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest_stats(batch=512):
    """Newest committed rocprofv3 --kernel-trace --stats summary of bench.py at this batch size (round, then version)."""
    def key(p):
        m = re.search(r"r(\d+)_bench_b\d+_kernel_stats(?:_v(\d+))?\.csv$", p)
        return (int(m.group(1)), int(m.group(2) or 0)) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_b{batch}_kernel_stats*.csv")), key=key)
    return files[-1] if files else None


def _commit_of(path):
    try:
        r = subprocess.run(["git", "-C", ROOT, "log", "-n", "1", "--format=%h", "--", path], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None
    except Exception:
        return None


def pick_dominant(stats_csv):
    """(name, row, skipped): the kernel with the largest TotalDurationNs that has a profile site; `skipped` lists larger rows
    without one (helpers launched several times per step with different sizes cannot be priced as ONE kernel)."""
    rows = sorted(csv.DictReader(open(stats_csv)), key=lambda r: -float(r["TotalDurationNs"]))
    skipped = []
    for r in rows:
        if r["Name"] in KERNELS:
            return r["Name"], r, skipped
        skipped.append({"name": r["Name"].split("(")[0], "pct": float(r["Percentage"]), "calls": int(r["Calls"])})
    return None, None, skipped


def time_site(eng, step_fn, site, steps=32):
    """Average duration (us) of the launch behind `site`, measured with HIP events inside `steps` real train steps."""
    check(eng.lib.eae_profile_enable(eng.ctx, site))
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    tot, emp, n = C.c_double(), C.c_double(), C.c_longlong()
    check(eng.lib.eae_profile_read2(eng.ctx, C.byref(tot), C.byref(emp), C.byref(n)))
    check(eng.lib.eae_profile_enable(eng.ctx, 0))
    k = max(n.value, 1)
    bracket_us, empty_us = 1e3 * tot.value / k, 1e3 * emp.value / k
    # a HIP event bracket around ONE launch also times the two event records; the empty bracket recorded right behind each timed
    # one measures them, the kernel's own duration -- what rocprofv3 reports -- is the difference
    return max(bracket_us - empty_us, 1e-3), bracket_us, empty_us, int(n.value)


def dominant_kernel_roofline(eng, step_fn, batch, hbm_peak_gbs, mfma_peak_tflops, steps=32):
    stats = newest_stats(512)
    name = row = None
    skipped = []
    if stats:
        name, row, skipped = pick_dominant(stats)
    if name is None:                       # no committed profile yet: the kernel that led the previous rounds' summaries
        name = "void wgrad_s2_kernel<64, 32, 16, 8, 1, 2, 1>(WgradArgs)"
    site, role, bpi, mf = KERNELS[name]
    us, bracket_us, empty_us, n = time_site(eng, step_fn, site, steps)
    byts = batch * bpi
    achieved = byts / (us * 1e-6) / 1e9
    out = {"kernel": name, "role": role, "bound": "hbm", "achieved": round(achieved, 1), "peak": hbm_peak_gbs, "unit": "GB/s",
           "frac": round(achieved / hbm_peak_gbs, 4), "traffic": None,
           "avg_launch_us": round(us, 2), "event_bracket_us": round(bracket_us, 2), "empty_bracket_us": round(empty_us, 2),
           "launches_timed": n, "algorithmic_bytes_per_launch": byts,
           "tflops": round(batch * mf * 1e6 / (us * 1e-6) / 1e12, 1),
           "mfma_frac": round(batch * mf * 1e6 / (us * 1e-6) / 1e12 / mfma_peak_tflops, 4),
           "selection": "largest TotalDurationNs among the single-launch-site kernels of the newest committed rocprofv3 summary"}
    # everything below is READ FROM COMMITTED FILES (rocprofv3 cannot run inside this process): kept apart from the live numbers
    prof = {}
    if stats and row is not None:
        prof = {"summary": os.path.relpath(stats, ROOT), "summary_commit": _commit_of(stats),
                "rocprof_avg_us": round(float(row["AverageNs"]) / 1e3, 2), "rocprof_pct_of_kernel_time": float(row["Percentage"]),
                "larger_rows_without_a_single_launch_site": skipped[:4]}
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_b512*.json")))
        if pm and batch == 512:
            k = json.load(open(pm[-1])).get("kernels", {}).get(name)
            if k:
                prof["pmc_file"] = os.path.relpath(pm[-1], ROOT)
                prof["pmc_commit"] = _commit_of(pm[-1])
                prof["pmc_traffic_bytes_per_launch"] = k["traffic_bytes"]
                out["traffic"] = k["traffic_bytes"]          # (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate --pmc passes
    out["from_committed_profile"] = prof
    return out
