#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/profile_round.sh) -> profiles/<tag>_pmc_traffic_b512.json
   usage: pmc_to_json.py <tag> <fetch_counter_collection.csv> <write_counter_collection.csv> [<mfma_counter_collection.csv>] [steps]

   traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch: gfx950's FETCH_SIZE reads half of wide coalesced streams
   (MI355X_MICROARCH.md, section HBM).  MFMA pass: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipe of a SIMD is busy, summed over the
   SIMDs), GRBM_GUI_ACTIVE (busy cycles, as reported: the sum over the 8 XCDs) and SQ_INSTS_VALU_MFMA_MOPS_BF16 (executed bf16 MFMA
   math ops / 512):  mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, ctr):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == ctr:
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


tag = sys.argv[1]
f = load(sys.argv[2], "FETCH_SIZE"); w = load(sys.argv[3], "WRITE_SIZE")
mf = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4].endswith(".csv") else None
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE "
               "SQ_INSTS_VALU_MFMA_MOPS_BF16 (separate passes, tools/profile_round.sh) of `bench.py --steps 5 --warmup 2 --no-cpu-baseline "
               "--no-roofline --no-configs`, B=512, averages per launch; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE "
               "reads half of wide coalesced streams, MI355X_MICROARCH.md section HBM); mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)",
       "kernels": {}}
busy = gui = mops = None
if mf:
    busy, gui, mops = load(mf, "SQ_VALU_MFMA_BUSY_CYCLES"), load(mf, "GRBM_GUI_ACTIVE"), load(mf, "SQ_INSTS_VALU_MFMA_MOPS_BF16")
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
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_b512.json")
json.dump(out, open(path, "w"), indent=1)
steps = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 7
ours = [v for k, v in out["kernels"].items() if not k.startswith(("void at::", "__amd_rocclr"))]
print(path, "HBM traffic per step (engine kernels): %.0f MB" % (sum(v["traffic_bytes"] * v["launches"] for v in ours) / steps / 1e6))
