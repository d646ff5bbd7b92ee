#!/bin/bash
# PMC passes over tools/inter_bench.py (one stage), summed per kernel name.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WHAT=${1:-mc}; KERN=${2:-k_mc_cells}
i=60
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 tools/inter_bench.py 64 $WHAT > gpurun_out/pmc_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$i/*/*counter_collection.csv")
if not f: print("no counter file for set: $set"); raise SystemExit
tot=collections.Counter(); n=0
for r in csv.DictReader(open(f[0])):
    if "$KERN" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
calls=len({r["Dispatch_Id"] for r in csv.DictReader(open(f[0])) if "$KERN" in r["Kernel_Name"]})
print(calls, {k:int(v/max(calls,1)) for k,v in tot.items()})
PY
done
