#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=${1:-512}; T=${2:-mix}
i=40
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$i
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --frames $F --steps 1 --warmup 0 --no-cpu-baseline --tiling $T > gpurun_out/pmc_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$i/*/*counter_collection.csv")
if not f: print("no counter file for set: $set"); raise SystemExit
tot=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "k_intra_level" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
print({k:int(v) for k,v in tot.items()})
PY
done
