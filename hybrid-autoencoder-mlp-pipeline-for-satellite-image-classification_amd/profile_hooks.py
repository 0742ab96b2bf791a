"""Per-kernel timing for bench.py's `roofline` objects.

The engine can bracket ONE launch site with HIP events on the stream that launch goes to (eae_profile_enable(ctx, site),
include/eae.h: site = EAE_PROF_SITE(layer, role), every forward / backward-data / weight-gradient launch of the eight conv layers).
`site_of(name)` maps the exact rocprofv3 kernel name to its site from the kernel's template arguments (kind, channel counts, source
and epilogue modes name the layer and the role whatever tile geometry was picked for the workload), `site_model(site, H, W)` gives
the site's algorithmic bytes / FLOPs per image (SURVEY.md 8d minimal-traffic model: every logical tensor the kernel must read or
write counted once, bf16 activations, fp32 input image; the <= 2.4 MB of weights are L2-resident and excluded).  bench.py picks the
site of the kernel with the largest TotalDurationNs in the newest committed rocprofv3 summary of the workload under profiles/ (or,
when there is none, the site with the longest live duration).
"""
from __future__ import annotations

import csv
import ctypes as C
import glob
import json
import math
import os
import re
import subprocess

import torch

from ._lib import check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENC_C = (3, 32, 64, 128, 256)
LAYER_NAMES = ("enc.conv1", "enc.conv2", "enc.conv3", "enc.conv4", "dec.deconv1", "dec.deconv2", "dec.deconv3", "dec.deconv4")
ROLE_NAMES = ("forward", "backward-data", "weight gradient")


def prof_site(layer, role):
    return 16 + 3 * layer + role          # EAE_PROF_SITE


def _ilog2(v):
    r = int(round(math.log2(v)))
    if 2 ** r != v:
        raise ValueError(v)
    return r


def site_of(name):
    """rocprofv3 kernel name -> launch site, or None for a kernel without one (helpers launched several times per step with
    different sizes cannot be priced as ONE kernel)."""
    m = re.match(r"(?:void )?(\w+)<([^>]*)>", name)
    if not m:
        return None
    fam, args = m.group(1), [int(a) for a in m.group(2).split(",")]
    if fam.endswith("_kernel_g"):          # the grouped twin of a kernel (csrc/eae_group.h): same template arguments, same launch site
        fam = fam[:-2]
    try:
        if fam in ("igemm_s2_kernel", "igemm2_s2_kernel", "igemm8_s2_kernel"):
            kind, cin, cout, src, epi = args[0], args[1], args[2], args[7], args[8]
            if kind == 0:          # conv kernel: a conv layer's forward, or a transposed-conv layer's backward-data
                return prof_site(_ilog2(cin // 32) + 1, 0) if src != 2 else prof_site(4 + _ilog2(256 // cout), 1)
            return prof_site(4 + _ilog2(256 // cin), 0) if src != 2 else prof_site(_ilog2(cin // 32), 1)
        if fam in ("wgrad_s2_kernel", "wgrad8_s2_kernel"):
            cs, smode, bmode = args[0], args[5], args[6]
            return prof_site(_ilog2(cs // 32), 2) if smode == 2 else prof_site(4 + _ilog2(256 // cs), 2)
        if fam == "edge_conv_kernel":
            return {(0, 0): prof_site(0, 0), (1, 1): prof_site(7, 1)}.get((args[0], args[1]))
        if fam == "edge_wgrad_kernel":
            return {(0, 2): prof_site(0, 2), (1, 1): prof_site(7, 2)}.get((args[0], args[1]))
        if fam == "deconv4_loss_kernel":
            return prof_site(7, 0)
    except (ValueError, IndexError):
        return None
    return None


def site_model(site, H=64, W=64):
    """(role text, algorithmic bytes per image, MFLOP per image) of a launch site for H x W inputs."""
    layer, role = divmod(site - 16, 3)
    x = 3 * H * W                                            # input image elements (fp32)

    def enc(i):                                              # elements of enc.conv(i+1)'s output map
        return ENC_C[i + 1] * (H >> (i + 1)) * (W >> (i + 1))

    def dec(i):                                              # elements of dec.deconv(i+1)'s output map, i = 0..2; dec(-1) = dec.fc's output
        return (128 >> i) * (H >> (3 - i)) * (W >> (3 - i)) if i >= 0 else 256 * (H >> 4) * (W >> 4)
    g4 = 4 * H * W                                           # deconv4's output gradient, NHWC4 bf16
    if layer < 4:
        big = x if layer == 0 else enc(layer - 1)            # the conv's input map
        small = enc(layer)
        bin_ = 4 * x if layer == 0 else 2 * big
        mflop = 2.0 * (H >> (layer + 1)) * (W >> (layer + 1)) * 9 * ENC_C[layer] * ENC_C[layer + 1] / 1e6
        byts = (bin_ + 2 * small,                            # forward: read the input map, write the raw output
                2 * 2 * small + 2 * big + 2 * big,           # backward-data: g, y of the output; y of the input (ReLU mask); write g of the input
                2 * 2 * small + bin_)[role]                  # weight gradient: g, y of the output, the input map
        what = ("", " (reads g, y of its output map and y of its input map for the ReLU mask, writes g of the input map)",
                " (reads g, y of its output map and its input map)")[role]
    else:
        i = layer - 4
        small = dec(i - 1)                                   # the transposed conv's input map
        cin = 256 >> i
        mflop = 2.0 * (H >> (4 - i)) * (W >> (4 - i)) * 9 * cin * (3 if i == 3 else cin // 2) / 1e6
        if i == 3:
            byts = (2 * small + 4 * x + 2 * g4,              # deconv4 + sigmoid + MSE + gradient: read u3 and the target, write g4
                    2 * g4 + 2 * small + 2 * small,          # backward-data: g4, y of the input map (mask), write its g
                    2 * g4 + 2 * small)[role]
            what = (" + sigmoid + MSE + its gradient", " (+ReLU mask, BN-backward sums)", "")[role]
        else:
            big = dec(i)
            byts = (2 * small + 2 * big,
                    2 * 2 * big + 2 * small + (2 * small if i > 0 else 0),
                    2 * small + 2 * 2 * big)[role]
            what = ("", " (reads g, y of its output map" + (", y of its input map for the ReLU mask" if i > 0 else "") + ", writes g of the input map)",
                    " (reads its input map and g, y of its output map)")[role]
    return f"{LAYER_NAMES[layer]} {ROLE_NAMES[role]}{what}", byts, mflop


def all_sites():
    return [prof_site(l, r) for l in range(8) for r in range(3) if not (l == 0 and r == 1)]


def newest_stats(tag="b512"):
    """Newest committed rocprofv3 --kernel-trace --stats summary of bench.py for this workload tag (round, then version)."""
    def key(p):
        m = re.search(r"r(\d+)_bench_\w+_kernel_stats(?:_v(\d+))?\.csv$", p)
        return (int(m.group(1)), int(m.group(2) or 0)) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_{tag}_kernel_stats*.csv")), key=key)
    return files[-1] if files else None


def _commit_of(path):
    """Commit the profile was TAKEN at: written into profiles/<round>_profile_meta.json when the files are generated (the GPU box has
    no git); fall back to the commit that added the file."""
    m = re.match(r"(r\d+)_", os.path.basename(path))
    if m:
        meta = os.path.join(ROOT, "profiles", f"{m.group(1)}_profile_meta.json")
        if os.path.exists(meta):
            try:
                return json.load(open(meta)).get("commit")
            except Exception:
                pass
    try:
        r = subprocess.run(["git", "-C", ROOT, "log", "-n", "1", "--format=%h", "--", path], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None
    except Exception:
        return None


def pick_dominant(stats_csv):
    """(name, row, skipped): the kernel with the largest TotalDurationNs that has a launch site; `skipped` lists larger rows
    without one.  gate_kernel rows (one spinning wave that waits for another stream) are not work and are left out."""
    rows = sorted(csv.DictReader(open(stats_csv)), key=lambda r: -float(r["TotalDurationNs"]))
    skipped = []
    for r in rows:
        if r["Name"].startswith(("gate_kernel", "void gate_kernel")):
            continue
        if site_of(r["Name"]) is not None:
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


def live_dominant_site(eng, step_fn, steps=6):
    best = (None, -1.0)
    for s in all_sites():
        us, _, _, n = time_site(eng, step_fn, s, steps)
        if n and us > best[1]:
            best = (s, us)
    return best[0]


def dominant_kernel_roofline(eng, step_fn, batch, hbm_peak_gbs, mfma_peak_tflops, steps=32, tag="b512", H=64, W=64):
    stats = newest_stats(tag)
    name = row = None
    skipped = []
    if stats:
        name, row, skipped = pick_dominant(stats)
    if name is not None:
        site = site_of(name)
        selection = "largest TotalDurationNs among the single-launch-site kernels of the newest committed rocprofv3 summary of this workload"
    else:                                  # no committed profile of this workload yet
        site = live_dominant_site(eng, step_fn)
        selection = "no committed rocprofv3 summary of this workload: the launch site with the longest live duration"
    role, bpi, mf = site_model(site, H, W)
    us, bracket_us, empty_us, n = time_site(eng, step_fn, site, steps)
    byts = batch * bpi
    achieved = byts / (us * 1e-6) / 1e9
    out = {"kernel": name, "site": site, "role": role, "bound": "hbm", "achieved": round(achieved, 1), "peak": hbm_peak_gbs, "unit": "GB/s",
           "frac": round(achieved / hbm_peak_gbs, 4), "traffic": None,
           "avg_launch_us": round(us, 2), "event_bracket_us": round(bracket_us, 2), "empty_bracket_us": round(empty_us, 2),
           "launches_timed": n, "algorithmic_bytes_per_launch": byts,
           "tflops": round(batch * mf * 1e6 / (us * 1e-6) / 1e12, 1),
           "mfma_frac": round(batch * mf * 1e6 / (us * 1e-6) / 1e12 / mfma_peak_tflops, 4),
           "selection": selection}
    # everything below is READ FROM COMMITTED FILES (rocprofv3 cannot run inside this process): kept apart from the live numbers
    prof = {}
    if stats and row is not None:
        prof = {"summary": os.path.relpath(stats, ROOT), "summary_commit": _commit_of(stats),
                "rocprof_avg_us": round(float(row["AverageNs"]) / 1e3, 2), "rocprof_pct_of_kernel_time": float(row["Percentage"]),
                "larger_rows_without_a_single_launch_site": skipped[:4]}
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{tag}*.json")))
        if pm:
            pj = json.load(open(pm[-1]))
            k = pj.get("kernels", {}).get(name)
            if k and pj.get("batch", 512 if tag == "b512" else None) == batch:
                prof["pmc_file"] = os.path.relpath(pm[-1], ROOT)
                prof["pmc_commit"] = pj.get("commit") or _commit_of(pm[-1])
                prof["pmc_traffic_bytes_per_launch"] = k["traffic_bytes"]
                out["traffic"] = k["traffic_bytes"]          # (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, separate --pmc passes
    out["from_committed_profile"] = prof
    return out
