#!/usr/bin/env python3
"""Per-kernel roofline tables (SURVEY.md 8d 'Reporting') from the committed rocprofv3 summaries of one round:
   profiles/<round>_bench_<tag>_kernel_stats.csv (durations) + profiles/<round>_pmc_traffic_<tag>.json (FETCH/WRITE_SIZE and the
   matrix-core busy counters), tag = b512 (BASELINE configs[2], the headline), c2 (configs[1]), c5fp8 / c5bf16 (configs[4]'s shape).
   Algorithmic bytes / FLOPs follow the minimal-traffic model of SURVEY.md 8d (eae_amd.profile_hooks.site_model for the conv layers):
   every logical tensor a kernel must read or write counted once (bf16 activations, fp32 input / partial sums), conv weights excluded
   (L2-resident), the latent projections' weights included (33.5 MB each at configs[4]'s shape).
   `gate_kernel` rows (one spinning wave waiting for another stream's progress word) are NOT work: they are left out of every
   percentage and of the per-step sums.
   usage: tools/kernel_report.py r03 > profiles/r03_per_kernel_roofline.md"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eae_amd import profile_hooks as PH      # noqa: E402

HBM, MFMA = 8000.0, 2500.0          # GB/s, TFLOP/s (dense bf16)
WL = {"b512": ("BASELINE configs[2] (headline): joint step, B=512, 64x64, L=64", 512, 64, 64),
      "c2": ("BASELINE configs[1]: reconstruction only, B=256, 64x64, L=64", 256, 64, 64),
      "c5fp8": ("BASELINE configs[4]'s per-GPU shape, fp8 GEMM operands: joint step, B=128, 256x256, L=256", 128, 256, 256),
      "c5bf16": ("the same shape through the bf16 kernels: joint step, B=128, 256x256, L=256", 128, 256, 256),
      "grid8": ("the notebook's batch size: 8 grid configurations x B=64 per GROUPED step (eae_group_train_step), 64x64, L=64 -- "
                "per launch = all 8 members; Adam / pack rows move 8 models' parameters", 512, 64, 64)}
GROUP = {"grid8": 8}


def ungroup(n):
    """kernel name of a grouped twin -> the single kernel's (site / role lookup)"""
    n = re.sub(r"_g(<|\()", r"\1", n)
    return re.sub(r"\(GroupPack<(\w+)>, int\)", r"(\1)", n)
HELPERS = ("bn_finalize", "bn_bwd_finalize", "reduce_slices", "fc_splitk_reduce", "head_kernel", "loss_finalize", "signal_kernel",
           "fp8_", "amax", "quant", "scale")


def fc_model(name, B, hw, L):
    """latent projections: (role, bytes per launch, MFLOP per launch)"""
    K = 256 * (hw // 16) ** 2
    m = re.match(r"(?:void )?(fc_nt_kernel|fc_tn_kernel)<(\d+), (\d+)>", name)
    if not m:
        return None
    fam, a, b = m.group(1), int(m.group(2)), int(m.group(3))
    w = 2 * K * L
    fl = 2.0 * K * L * B / 1e6
    if fam == "fc_nt_kernel":
        role = {(1, 0): "enc.fc forward (split-K partials)", (3, 1): "dec.fc forward", (0, 0): "dec.fc backward-data (split-K partials)",
                (3, 2): "enc.fc backward-data (+mask, sums)"}.get((a, b))
        if role is None:
            return None
        byts = {(1, 0): B * 2 * K + w, (3, 1): B * (4 * L + 2 * K) + w, (0, 0): B * 2 * K + w, (3, 2): B * (4 * L + 4 * K) + w}[(a, b)]
        return role, byts, fl
    role = {(0, 3): "dec.fc weight gradient", (3, 1): "enc.fc weight gradient"}.get((a, b))
    if role is None:
        return None
    return role, B * (2 * K + 4 * L) + 4 * K * L, fl


def n_params(hw, L, classes=10):
    K = 256 * (hw // 16) ** 2
    enc = sum(co * ci * 9 + co + 2 * co for ci, co in ((3, 32), (32, 64), (64, 128), (128, 256)))
    dec = sum(ci * co * 9 + co + (2 * co if co != 3 else 0) for ci, co in ((256, 128), (128, 64), (64, 32), (32, 3)))
    return enc + dec + K * L + L + L * K + K + L * 128 + 128 + 128 * classes + classes


def table(rnd, tag):
    stats = os.path.join(ROOT, "profiles", f"{rnd}_bench_{tag}_kernel_stats.csv")
    pmc_path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{tag}.json")
    if not os.path.exists(stats):
        return
    desc, B, hw, L = WL[tag]
    rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
    pmc = json.load(open(pmc_path))["kernels"] if os.path.exists(pmc_path) else {}
    # stand-alone durations per launch site (tools/kbench_sites.py; the fp8 workload shows its bf16 twins' numbers)
    sa_path = os.path.join(ROOT, "profiles", f"{rnd}_standalone_{'c5bf16' if tag == 'c5fp8' else tag}.json")
    alone = {int(k): v for k, v in json.load(open(sa_path))["us"].items()} if os.path.exists(sa_path) else {}
    work = {n: r for n, r in rows.items() if "gate_kernel" not in n}
    tot = sum(float(r["TotalDurationNs"]) for r in work.values())
    print(f"\n## {desc}\n\n`{os.path.relpath(stats, ROOT)}`" + (f" + `{os.path.relpath(pmc_path, ROOT)}`" if pmc else "") + "\n")
    print("| kernel | role | calls | µs in the step | µs alone | % of kernel time | alg MB | GB/s | HBM frac | HBM frac alone | TFLOP/s | MFMA frac | PMC MB | PMC/alg | MFMA busy |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    lines = []
    ng = GROUP.get(tag, 1)
    for n0, r in work.items():
        n = ungroup(n0)
        us = float(r["AverageNs"]) / 1e3
        pct = 100.0 * float(r["TotalDurationNs"]) / tot
        site = PH.site_of(n)
        if site is not None:
            role, bpi, mf = PH.site_model(site, hw, hw)
            byts, fl = B * bpi, B * mf
        else:
            fm = fc_model(n, B, hw, L)
            if fm:
                role, byts, fl = fm
            elif n.startswith(("adam_kernel", "void adam_kernel", "adam_dyn_kernel", "void adam_dyn_kernel")):
                role, byts, fl = "Adam over the flat arenas (p, g, m, v read; p, m, v write)", ng * 7 * 4 * n_params(hw, L), 0.0
            elif "pack_all_kernel" in n:
                role, byts, fl = "fp32 master weights -> bf16 kernel layouts", ng * (4 + 2 * 2) * n_params(hw, L), 0.0
            else:
                continue
        k = pmc.get(n0, {})
        tr, mu = k.get("traffic_bytes"), k.get("mfma_util")
        short = n.split("(")[0].replace("void ", "")
        al = alone.get(site) if site is not None else None
        als = f"{al:.1f}" if al is not None else "–"
        if byts is None:
            lines.append((pct, f"| `{short}` | {role} | {r['Calls']} | {us:.1f} | {als} | {pct:.1f} | – | – | – | – | – | – | {tr / 1e6 if tr else float('nan'):.1f} | – | – |"))
            continue
        gbs = byts / (us * 1e-6) / 1e9
        tf = fl * 1e6 / (us * 1e-6) / 1e12
        fa = f"{byts / (al * 1e-6) / 1e9 / HBM:.2f}" if al else "–"
        lines.append((pct, f"| `{short}` | {role.split(' (')[0]} | {r['Calls']} | {us:.1f} | {als} | {pct:.1f} | {byts / 1e6:.1f} | {gbs:.0f} | {gbs / HBM:.2f} | {fa} | {tf:.0f} | "
                           f"{tf / MFMA:.3f} | {tr / 1e6 if tr else float('nan'):.1f} | {tr / byts if tr else float('nan'):.2f} | "
                           f"{mu if mu is not None else float('nan'):.3f} |"))
    for _, l in sorted(lines, key=lambda t: -t[0]):
        print(l)
    small = [(n.split("(")[0].replace("void ", ""), float(r["AverageNs"]) / 1e3, int(r["Calls"]), 100.0 * float(r["TotalDurationNs"]) / tot)
             for n, r in work.items() if PH.site_of(ungroup(n)) is None and any(k in n for k in HELPERS)]
    if small:
        print("\nHelpers (latency-bound, no meaningful roofline; average per launch, share of kernel time): " +
              "; ".join(f"`{n}` {us:.1f} µs × {c} ({p:.1f} %)" for n, us, c, p in sorted(small, key=lambda t: -t[3])))
    gates = [(float(r["AverageNs"]) / 1e3, int(r["Calls"])) for n, r in rows.items() if "gate_kernel" in n]
    if gates:
        print(f"\n`gate_kernel` (excluded above): {gates[0][1]} launches, {gates[0][0]:.1f} µs average of one wave spinning.")
    # steps = launches of a once-per-step kernel (the fp8 calibration adds gradient steps without an optimizer step)
    calls = [int(r["Calls"]) for n, r in rows.items() if "deconv4_loss_kernel" in n] or [int(r["Calls"]) for n, r in rows.items() if "adam" in n and "kernel" in n]
    if calls:
        steps = max(calls)
        line = f"\nSum of kernel durations per step (all streams, gates excluded): {tot / steps / 1e3:.0f} µs over {steps} steps"
        if pmc:
            pj = json.load(open(pmc_path))
            if "hbm_traffic_bytes_per_step" in pj:
                line += f"; HBM traffic per step by the PMC counters: {pj['hbm_traffic_bytes_per_step'] / 1e6:.0f} MB"
        print(line + ".")


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    meta_p = os.path.join(ROOT, "profiles", f"{rnd}_profile_meta.json")
    meta = json.load(open(meta_p)) if os.path.exists(meta_p) else {}
    print(f"# Per-kernel roofline, one MI355X, round {rnd[1:]} (library built from commit `{meta.get('commit', '?')}`)\n")
    print("Durations are rocprofv3 averages inside real train steps (they include the command processor's ≈2–3 µs per dispatch and, for the "
          "backward kernels, contention with the weight-gradient streams that run beside them). `alg MB` = algorithmic bytes per launch "
          "(SURVEY §8d model), `HBM frac` = alg bytes / time / 8 TB/s, `MFMA frac` = FLOPs / time / 2.5 PFLOP/s (bf16 dense peak, also for "
          "the fp8 kernels' rows), `PMC MB` = (2·FETCH_SIZE + WRITE_SIZE) per launch, `MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / "
          "(GRBM_GUI_ACTIVE/8 · 1024 SIMDs): the fraction of the kernel's lifetime in which a SIMD's matrix pipe was busy, averaged over "
          "all SIMDs (separate `--pmc` pass). `% of kernel time` excludes `gate_kernel`. `µs alone` = the same launch site through the "
          "per-op C ABI, back to back on an otherwise idle GPU (`tools/kbench_sites.py`; weight-gradient rows include their slice-reduction "
          "launch, so they can exceed the in-step kernel alone; the fp8 table shows its bf16 twins): the gap to `µs in the step` is "
          "contention -- side streams beside the backward-data chain -- not kernel quality.")
    for tag in ("b512", "c2", "c5fp8", "c5bf16", "grid8"):
        table(rnd, tag)


if __name__ == "__main__":
    main()
