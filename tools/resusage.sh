#!/bin/bash
# usage: tools_resusage.sh file.hip  -> per-kernel VGPR/AGPR/SGPR/spill/occupancy/LDS table
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c "$f" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        name=t.split(":",1)[1].strip()
        try: name=subprocess.run(["c++filt",name],capture_output=True,text=True).stdout.strip()
        except Exception: pass
        cur={"name":re.sub(r"\(.*","",name)}; rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
for r in rows:
    print("%-60s V=%-4s A=%-4s S=%-4s spillV=%-3s occ=%-2s LDS=%s" % (r["name"][:60], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
