#!/bin/bash
# usage: tools/kres.sh <file.hip>  -> kernel resource usage table (VGPRs, AGPRs, scratch, LDS, occupancy)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur={"name":subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()[:90]}; rows.append(cur); continue
    for k in ("VGPRs","AGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","LDS Size \[bytes/block\]","SGPRs"):
        m=re.search(k+r": (\d+)",l)
        if m and cur is not None: cur[k.split(" ")[0]]=m.group(1)
for r in rows:
    print(f"{r.get(\"VGPRs\",\"?\"):>4} vgpr {r.get(\"AGPRs\",\"?\"):>3} agpr {r.get(\"ScratchSize\",\"?\"):>4} scr {r.get(\"LDS\",\"?\"):>6} lds occ {r.get(\"Occupancy\",\"?\")}  {r[\"name\"]}")
'
