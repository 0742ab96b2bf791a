#!/usr/bin/env python3
"""usage: tools/kres.py <file.hip>  -> kernel resource usage table (VGPRs, scratch, LDS, occupancy) via hipcc remarks"""
import re
import subprocess
import sys

r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", sys.argv[1], "-o", "/tmp/kres.o",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur, rows = None, []
for l in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()[:100]}
        rows.append(cur)
        continue
    for k, key in (("VGPRs", "v"), ("AGPRs", "a"), ("ScratchSize", "s"), ("Occupancy", "o"), ("LDS Size", "l")):
        m = re.search(re.escape(k) + r"[^:]*: (\d+)", l)
        if m and cur is not None and key not in cur:
            cur[key] = m.group(1)
for x in rows:
    print("%4s vgpr %3s agpr %4s scratch %6s lds occ %s  %s" % (x.get("v"), x.get("a"), x.get("s"), x.get("l"), x.get("o"), x["name"]))
