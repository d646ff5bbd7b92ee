#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=${1:-512}; T=${2:-4}
i=20
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$i
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --frames $F --steps 1 --warmup 0 --no-cpu-baseline --tiling $T > gpurun_out/pmc_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$i/*/*counter_collection.csv")
if not f: print("no counter file for set $i: $set"); raise SystemExit
tot=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0][-28:]
    tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in tot.items(): print(k,{a:int(b) for a,b in v.items()})
PY
done
