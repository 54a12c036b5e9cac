#!/bin/bash
# usage: res.sh file.hip  -> per kernel: VGPRs, AGPRs, scratch, occupancy, spills
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/protoasnet_amd/csrc -c $1 -o /tmp/w/res.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark: (?:\S+ )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", line)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=="Function Name":
        cur={"name":v.replace("_ZN4pasn","")[:56]}; rows.append(cur)
    elif cur is not None: cur[k.split(" [")[0]]=v
for r in rows:
    print("%-56s V %3s A %3s scratch %4s occ %s sgpr-spill %3s vgpr-spill %3s"%(r["name"],r.get("VGPRs"),r.get("AGPRs"),r.get("ScratchSize"),r.get("Occupancy"),r.get("SGPRs Spill"),r.get("VGPRs Spill")))
'
