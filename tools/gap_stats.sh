#!/bin/bash
# duration of the level launches and the idle gap between consecutive ones (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=${1:-512}
rm -rf gpurun_out/gap
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python3 bench.py --frames $F --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/gap.log 2>&1
python3 - <<PY
import csv,glob,statistics as st
f=glob.glob("gpurun_out/gap/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "k_intra_level" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[len(rows)//2:]   # the timed step
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
g=[int(rows[i+1]["Start_Timestamp"])-int(rows[i]["End_Timestamp"]) for i in range(len(rows)-1)]
print("launches",len(rows),"duration us mean/median",round(st.mean(d)/1e3,2),round(st.median(d)/1e3,2),"gap us mean/median",round(st.mean(g)/1e3,2),round(st.median(g)/1e3,2))
print("sum duration ms",round(sum(d)/1e6,2),"sum gap ms",round(sum(g)/1e6,2))
PY
