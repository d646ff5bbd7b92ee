#!/bin/bash
# per-tiling instruction mix of k_intra_level: one PMC pass per tiling (no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=${1:-128}
for T in 4 8 16 32 mix; do
  rm -rf gpurun_out/pmc6_$T
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc6_$T -- python3 bench.py --frames $F --steps 1 --warmup 0 --no-cpu-baseline --tiling $T > gpurun_out/pmc6_$T.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc6_$T/*/*counter_collection.csv")
if not f: print("no counter file for tiling $T"); raise SystemExit
tot=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "k_intra_level" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
w=tot["SQ_WAVES"]
print("tiling $T", "waves", int(w), {k: round(v/w,1) for k,v in tot.items() if k!="SQ_WAVES"})
PY
done
