#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, one counter per pass (MI355X_MICROARCH.md:
# FETCH_SIZE and WRITE_SIZE cannot share a pass), kernel-trace only.  Writes profiles/<tag>_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=${1:-512}; TAG=${2:-r01}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/traffic_$c
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/traffic_$c -- python3 bench.py --frames $F --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/traffic_$c.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
out={"frames":$F,"steps":1,"note":"rocprofv3 --pmc, separate passes; values in KB as reported; k_intra_level<true> only"}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob("gpurun_out/traffic_%s/*/*counter_collection.csv"%c)[0]
    tot=0.0;n=0
    for r in csv.DictReader(open(f)):
        if "k_intra_level" in r["Kernel_Name"] and r["Counter_Name"]==c:
            tot+=float(r["Counter_Value"]); n+=1
    out[c+"_KB"]=tot; out["launches"]=n
json.dump(out,open("gpurun_out/%s_traffic.json"%"$TAG","w"),indent=1)
print(out)
PY
