#!/usr/bin/env python3
"""After tools/profile_round.sh ran on the GPU box and gpurun merged gpurun_out/prof_<round>/ back:
       tools/profile_collect.py <round> <commit> [workload ...]
   copies each workload's rocprofv3 kernel-stats summary to profiles/<round>_bench_<tag>_kernel_stats.csv, turns its three --pmc
   passes into profiles/<round>_pmc_traffic_<tag>.json and writes profiles/<round>_profile_meta.json with the commit the library was
   built from (the GPU box has no git: bench.py reads the commit from this file).  tag: c3 -> b512, the others keep their name.

   traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch: gfx950's FETCH_SIZE reads half of wide coalesced streams
   (MI355X_MICROARCH.md, section HBM).  MFMA pass: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe of a SIMD is busy, summed over the
   SIMDs), GRBM_GUI_ACTIVE (busy cycles, as reported: the sum over the 8 XCDs) and SQ_INSTS_VALU_MFMA_MOPS_BF16 (executed bf16 MFMA
   math ops / 512):  mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH = {"c3": 512, "c2": 256, "c5fp8": 128, "c5bf16": 128, "grid8": 512}
TAG = {"c3": "b512"}


def load(path, ctr):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


def one(path_glob):
    f = glob.glob(path_glob, recursive=True)
    return f[0] if f else None


def main():
    rnd, commit = sys.argv[1], sys.argv[2]
    wls = sys.argv[3:] or ["c3", "c2", "c5fp8", "c5bf16", "grid8"]
    meta = {"commit": commit, "collected": time.strftime("%Y-%m-%d"), "tool": "tools/profile_round.sh + tools/profile_collect.py", "workloads": {}}
    for wl in wls:
        d = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}", wl)
        tag = TAG.get(wl, wl)
        st = one(os.path.join(d, "trace", "**", "*kernel_stats.csv"))
        if not st:
            print("no kernel stats for", wl); continue
        dst = os.path.join(ROOT, "profiles", f"{rnd}_bench_{tag}_kernel_stats.csv")
        shutil.copy(st, dst)
        run = open(os.path.join(d, "run.txt")).read().strip() if os.path.exists(os.path.join(d, "run.txt")) else ""
        meta["workloads"][wl] = {"kernel_stats": os.path.relpath(dst, ROOT), "batch": BATCH[wl], "run": run,
                                 "command": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --workload {wl} --no-cpu-baseline --no-roofline --no-configs"}
        sa = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}", f"standalone_{tag}.json")      # tools/kbench_sites.py
        if os.path.exists(sa):
            sdst = os.path.join(ROOT, "profiles", f"{rnd}_standalone_{tag}.json")
            shutil.copy(sa, sdst)
            meta["workloads"][wl]["standalone"] = os.path.relpath(sdst, ROOT)
        fp = one(os.path.join(d, "FETCH_SIZE", "**", "*counter_collection.csv"))
        wp = one(os.path.join(d, "WRITE_SIZE", "**", "*counter_collection.csv"))
        mp = one(os.path.join(d, "MFMA", "**", "*counter_collection.csv"))
        if not (fp and wp):
            print("no PMC passes for", wl); continue
        f, w = load(fp, "FETCH_SIZE"), load(wp, "WRITE_SIZE")
        busy = gui = mops = None
        if mp:
            busy, gui, mops = load(mp, "SQ_VALU_MFMA_BUSY_CYCLES"), load(mp, "GRBM_GUI_ACTIVE"), load(mp, "SQ_INSTS_VALU_MFMA_MOPS_BF16")
        out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
                       "SQ_INSTS_VALU_MFMA_MOPS_BF16 (separate passes, tools/profile_round.sh, EAE_FORK_EVENTS=1) of `bench.py --workload " + wl +
                       " --no-cpu-baseline --no-roofline --no-configs`, averages per launch; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                       "(gfx950 FETCH_SIZE reads half of wide coalesced streams, MI355X_MICROARCH.md section HBM); "
                       "mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)",
               "commit": commit, "workload": wl, "batch": BATCH[wl], "kernels": {}}
        for k in f:
            fk = f[k][0] / f[k][1]; wk = w[k][0] / w[k][1] if k in w else 0.0
            e = {"FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1), "traffic_bytes": int((2 * fk + wk) * 1024), "launches": f[k][1]}
            if busy is not None and k in busy and k in gui and gui[k][0] > 0:
                b, g = busy[k][0] / busy[k][1], gui[k][0] / gui[k][1]
                e["mfma_busy_cycles"] = round(b, 0)
                e["gui_active_cycles_sum8"] = round(g, 0)
                e["mfma_util"] = round(b / (g / 8.0 * 1024.0), 4)
                if k in mops:
                    e["mfma_flop_bf16"] = int(mops[k][0] / mops[k][1] * 512)
            out["kernels"][k] = e
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{tag}.json")
        json.dump(out, open(path, "w"), indent=1)
        # steps seen by the counter pass = launches of a once-per-step kernel (fp8 calibration adds gradient steps of its own)
        steps = max([v["launches"] for k, v in out["kernels"].items() if "deconv4_loss_kernel" in k] or [7])
        ours = [v for k, v in out["kernels"].items() if not k.startswith(("void at::", "__amd_rocclr"))]
        per_step = sum(v["traffic_bytes"] * v["launches"] for v in ours) / steps
        out["hbm_traffic_bytes_per_step"] = int(per_step)
        json.dump(out, open(path, "w"), indent=1)
        meta["workloads"][wl]["pmc"] = os.path.relpath(path, ROOT)
        print(wl, path, "HBM traffic per step (engine kernels): %.0f MB" % (per_step / 1e6))
    json.dump(meta, open(os.path.join(ROOT, "profiles", f"{rnd}_profile_meta.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
