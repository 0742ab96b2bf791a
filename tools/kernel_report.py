#!/usr/bin/env python3
"""Per-kernel roofline table (SURVEY.md 8d 'Reporting') from the committed rocprofv3 summaries:
   newest profiles/r<NN>_bench_b512_kernel_stats_v<N>.csv (durations) + newest profiles/r<NN>_pmc_traffic_b512.json (FETCH/WRITE_SIZE
   and the matrix-core busy counters).
   Algorithmic bytes / FLOPs per image follow the minimal-traffic model of SURVEY.md 8d: every logical tensor a kernel must read
   or write counted once (bf16 activations, fp32 input / partial sums), weights excluded (L2-resident, <= 0.6 MB).
   usage: tools/kernel_report.py [stats.csv] > profiles/r02_per_kernel_roofline.md"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 512
HBM, MFMA = 8000.0, 2500.0          # GB/s, TFLOP/s (dense bf16)
X, Y1, Y2, Y3, Y4 = 3 * 64 * 64, 32 * 32 * 32, 16 * 16 * 64, 8 * 8 * 128, 4 * 4 * 256      # elements per image
MF = 2.0 * 256 * 288 * 64 / 1e6     # 9.44 MFLOP: every 3x3 s2 layer between 32 and 256 channels does the same work per image
EDGE_MF = 2.0 * 1024 * 27 * 32 / 1e6
# name fragment -> (what, bytes per image, MFLOP per image)
T = [
    ("edge_conv_kernel<0, 0>", "enc.conv1 forward", 4 * X + 2 * Y1, EDGE_MF),
    ("igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 1, 0>", "enc.conv2 forward", 2 * Y1 + 2 * Y2, MF),
    ("igemm_s2_kernel<0, 64, 128, 64, 8, 8, 2, 1, 0>", "enc.conv3 forward", 2 * Y2 + 2 * Y3, MF),
    ("s2_kernel<0, 128, 256, 64, 4, 4, 8, 1, 0", "enc.conv4 forward (wave-specialised kernel)", 2 * Y3 + 2 * Y4, MF),
    ("fc_nt_kernel<1, 0>", "enc.fc forward (split-K partials)", 2 * Y4 + 32 * 64 * 4, 2.0 * 4096 * 64 / 1e6),
    ("fc_nt_kernel<3, 1>", "dec.fc forward", 64 * 4 + 2 * Y4, 2.0 * 4096 * 64 / 1e6),
    ("s2_kernel<1, 256, 128, 64, 4, 4, 4, 0, 0", "dec.deconv1 forward (wave-specialised kernel)", 2 * Y4 + 2 * Y3, MF),
    ("igemm_s2_kernel<1, 128, 64, 64, 8, 8, 1, 1, 0>", "dec.deconv2 forward", 2 * Y3 + 2 * Y2, MF),
    ("igemm_s2_kernel<1, 64, 32, 32, 16, 8, 1, 1, 0>", "dec.deconv3 forward", 2 * Y2 + 2 * Y1, MF),
    ("deconv4_loss_kernel<1>", "dec.deconv4 + sigmoid + MSE + its gradient", 2 * Y1 + 4 * X + 2 * 4 * 64 * 64, EDGE_MF),
    ("edge_conv_kernel<1, 1>", "deconv4 backward-data (+ReLU mask, BN-bwd sums)", 2 * 4 * 64 * 64 + 2 * Y1 + 2 * Y1, EDGE_MF),
    ("igemm_s2_kernel<0, 32, 64, 64, 16, 8, 1, 2, 1>", "deconv3 backward-data", 2 * 2 * Y1 + 2 * Y2 + 2 * Y2, MF),
    ("igemm_s2_kernel<0, 64, 128, 64, 8, 8, 2, 2, 1>", "deconv2 backward-data", 2 * 2 * Y2 + 2 * Y3 + 2 * Y3, MF),
    ("igemm_s2_kernel<0, 128, 256, 64, 4, 4, 8, 2, 2>", "deconv1 backward-data", 2 * 2 * Y3 + 2 * Y4, MF),
    ("fc_nt_kernel<0, 0>", "dec.fc backward-data (split-K partials)", 2 * Y4 + 32 * 64 * 4, 2.0 * 4096 * 64 / 1e6),
    ("fc_nt_kernel<3, 2>", "enc.fc backward-data (+mask, sums)", 64 * 4 + 2 * Y4 + 2 * Y4, 2.0 * 4096 * 64 / 1e6),
    ("igemm_s2_kernel<1, 256, 128, 64, 4, 4, 4, 2, 1>", "conv4 backward-data", 2 * 2 * Y4 + 2 * Y3 + 2 * Y3, MF),
    ("igemm_s2_kernel<1, 128, 64, 64, 8, 8, 1, 2, 1>", "conv3 backward-data", 2 * 2 * Y3 + 2 * Y2 + 2 * Y2, MF),
    ("igemm_s2_kernel<1, 64, 32, 32, 16, 8, 1, 2, 1>", "conv2 backward-data", 2 * 2 * Y2 + 2 * Y1 + 2 * Y1, MF),
    ("edge_wgrad_kernel<1, 1>", "deconv4 weight gradient", 2 * 4 * 64 * 64 + 2 * Y1, EDGE_MF),
    ("edge_wgrad_kernel<0, 2>", "conv1 weight gradient", 4 * X + 2 * 2 * Y1, EDGE_MF),
    ("wgrad_s2_kernel<64, 32, 16, 8, 1, 1, 2>", "deconv3 weight gradient", 2 * Y2 + 2 * 2 * Y1, MF),
    ("wgrad_s2_kernel<128, 64, 8, 8, 2, 1, 2>", "deconv2 weight gradient", 2 * Y3 + 2 * 2 * Y2, MF),
    ("wgrad_s2_kernel<256, 128, 4, 4, 8, 0, 2>", "deconv1 weight gradient", 2 * Y4 + 2 * 2 * Y3, MF),
    ("wgrad_s2_kernel<256, 128, 4, 4, 8, 2, 1>", "conv4 weight gradient", 2 * 2 * Y4 + 2 * Y3, MF),
    ("wgrad_s2_kernel<128, 64, 8, 8, 2, 2, 1>", "conv3 weight gradient", 2 * 2 * Y3 + 2 * Y2, MF),
    ("wgrad_s2_kernel<64, 32, 16, 8, 1, 2, 1>", "conv2 weight gradient", 2 * 2 * Y2 + 2 * Y1, MF),
    ("fc_tn_kernel<0, 3>", "dec.fc weight gradient", 2 * Y4 + 64 * 4, 2.0 * 4096 * 64 / 1e6),
    ("fc_tn_kernel<3, 1>", "enc.fc weight gradient", 64 * 4 + 2 * Y4, 2.0 * 4096 * 64 / 1e6),
]
P_ARENA = 1316048 * 4
PER_LAUNCH = [      # kernels whose traffic does not scale with the batch: bytes per launch
    ("adam_kernel", "Adam over the flat arenas (p,g,m,v read; p,m,v write)", 7 * P_ARENA),
    ("pack_all_kernel", "fp32 master weights -> bf16 kernel layouts", P_ARENA + 2 * 2 * 1310000),
]


def newest(pattern, key):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=key)
    return files[-1] if files else None


def main():
    import re

    def skey(p):
        m = re.search(r"r(\d+)_bench_b512_kernel_stats(?:_v(\d+))?\.csv$", p)
        return (int(m.group(1)), int(m.group(2) or 0))
    stats = sys.argv[1] if len(sys.argv) > 1 else newest("r*_bench_b512_kernel_stats*.csv", skey)
    pmc_path = newest("r*_pmc_traffic_b512.json", lambda p: p)
    rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
    pmc = json.load(open(pmc_path))["kernels"]
    print(f"# Per-kernel roofline, B={B}, one MI355X (`{os.path.relpath(stats, ROOT)}` + `{os.path.relpath(pmc_path, ROOT)}`)\n")
    print("Durations are rocprofv3 averages inside real train steps (they include the command processor's ≈2–3 µs per dispatch and, for the "
          "backward kernels, contention with the weight-gradient streams that run beside them). `alg MB` = algorithmic bytes per launch "
          "(SURVEY §8d model), `HBM frac` = alg bytes / time / 8 TB/s, `MFMA frac` = FLOPs / time / 2.5 PFLOP/s, `PMC MB` = "
          "(2·FETCH_SIZE + WRITE_SIZE) per launch, `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 · 1024 SIMDs): the fraction "
          "of the kernel's lifetime in which a SIMD's matrix pipe was busy, averaged over all SIMDs (separate `--pmc` pass).\n")
    print("| kernel | role | µs | alg MB | GB/s | HBM frac | TFLOP/s | MFMA frac | PMC MB | PMC/alg | MFMA busy |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")

    def find(frag):
        for n, r in rows.items():
            if frag in n:
                return n, r
        return None, None

    for frag, what, bpi, mf in T:
        n, r = find(frag)
        if r is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        mb = B * bpi / 1e6
        gbs = B * bpi / (us * 1e-6) / 1e9
        tf = B * mf * 1e6 / (us * 1e-6) / 1e12
        k = pmc.get(n, {})
        tr = k.get("traffic_bytes")
        mu = k.get("mfma_util")
        short = n.split("(")[0].replace("void ", "")
        print(f"| `{short}` | {what} | {us:.1f} | {mb:.1f} | {gbs:.0f} | {gbs / HBM:.2f} | {tf:.0f} | {tf / MFMA:.3f} | "
              f"{tr / 1e6 if tr else float('nan'):.1f} | {tr / 1e6 / mb if tr else float('nan'):.2f} | {mu if mu is not None else float('nan'):.3f} |")
    for frag, what, byts in PER_LAUNCH:
        n, r = find(frag)
        if r is None:
            continue
        us = float(r["AverageNs"]) / 1e3
        gbs = byts / (us * 1e-6) / 1e9
        tr = pmc.get(n, {}).get("traffic_bytes")
        print(f"| `{frag}` | {what} | {us:.1f} | {byts / 1e6:.1f} | {gbs:.0f} | {gbs / HBM:.2f} | – | – | {tr / 1e6 if tr else float('nan'):.1f} | – | – |")
    small = [(n, float(r["AverageNs"]) / 1e3, int(r["Calls"])) for n, r in rows.items()
             if any(k in n for k in ("bn_finalize", "bn_bwd_finalize", "reduce_slices", "fc_splitk_reduce", "head_kernel", "loss_finalize",
                                     "gate_kernel", "signal_kernel"))]
    print("\nLatency-bound helpers (no meaningful roofline; average per launch; `gate_kernel` = a side stream waiting for the main "
          "stream's progress word, one wave): " + "; ".join(f"`{n.split('(')[0]}` {us:.1f} µs" for n, us, _ in sorted(small)))
    tot = sum(float(r["TotalDurationNs"]) for n, r in rows.items() if "gate_kernel" not in n)
    calls = [int(r["Calls"]) for n, r in rows.items() if n.startswith("adam_kernel")]
    if calls:
        print(f"\nSum of kernel durations per step (all streams, gates excluded): {tot / calls[0] / 1e3:.0f} µs.")


if __name__ == "__main__":
    main()
