#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE counter_collection CSVs (two separate passes) -> profiles/r01_pmc_traffic_b512.json
   usage: pmc_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <n_steps_profiled>"""
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


f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of `bench.py --steps 5 --warmup 2 "
               "--no-cpu-baseline --no-roofline`, B=512, average per launch in KB as reported; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
               "(gfx950 FETCH_SIZE reads half of wide coalesced streams, MI355X_MICROARCH.md section HBM)", "kernels": {}}
for k in f:
    fk = f[k][0] / f[k][1]; wk = w[k][0] / w[k][1] if k in w else 0.0
    out["kernels"][k] = {"FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1), "traffic_bytes": int((2 * fk + wk) * 1024), "launches": f[k][1]}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_b512.json"), "w"), indent=1)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
ours = [v for k, v in out["kernels"].items() if not k.startswith(("void at::", "__amd_rocclr"))]
print("HBM traffic per step (engine kernels): %.0f MB" % (sum(v["traffic_bytes"] * v["launches"] for v in ours) / steps / 1e6))
