#!/usr/bin/env python3
"""Per-stream timeline of one train step from a rocprofv3 --kernel-trace CSV (diagnostic).  usage: timeline.py <kernel_trace.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
k = len(adam) // 2
step = rows[adam[k - 1] + 1: adam[k] + 1]
t0 = step[0]["s"]


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:46]


streams = sorted(set(r["Stream_Id"] for r in step))
print(f"step of {len(step)} kernels, {(step[-1]['e'] - t0) / 1e3:.1f} us; streams {streams}")
for sid in streams:
    ks = [r for r in step if r["Stream_Id"] == sid]
    busy = sum(r["e"] - r["s"] for r in ks) / 1e3
    print(f"--- stream {sid}: {len(ks)} kernels, busy {busy:.1f} us, first start {(ks[0]['s'] - t0) / 1e3:.1f}, last end {(ks[-1]['e'] - t0) / 1e3:.1f}")
    prev = None
    for r in ks:
        gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
        print(f"   {(r['s'] - t0) / 1e3:8.1f} +{(r['e'] - r['s']) / 1e3:6.1f}  gap {gap:6.1f}  {short(r['Kernel_Name'])}")
        prev = r["e"]
