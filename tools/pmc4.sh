#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=${1:-256}; T=${2:-mix}
rm -rf gpurun_out/pmc_q
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_q -- python3 bench.py --frames $F --steps 1 --warmup 0 --no-cpu-baseline --tiling $T > gpurun_out/pmc_q.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_q/*/*counter_collection.csv")
tot=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].split("(")[0][-24:]
    tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in tot.items():
    if "intra" in k or "convert" in k:
        w=v["SQ_WAVES"]; print(k, "waves %d valu/wave %.0f salu/wave %.0f lds/wave %.0f wavecyc/wave %.0f wait%% %.0f active%% %.0f"%(w,v["SQ_INSTS_VALU"]/w,v["SQ_INSTS_SALU"]/w,v["SQ_INSTS_LDS"]/w,v["SQ_WAVE_CYCLES"]/w,100*v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"],100*v["SQ_ACTIVE_INST_ANY"]/v["SQ_WAVE_CYCLES"]))
PY
