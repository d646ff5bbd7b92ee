#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F=${1:-256}
rocprofv3 -L 2>/dev/null | grep -o "SQ[C]*_[A-Z_0-9]*" | sort -u | grep -i "ifetch\|icache\|ICACHE\|LDS_\|SQ_WAIT\|TCP_\|DCACHE" | tr '\n' ' ' > gpurun_out/counters_list.txt
i=10
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_$i
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$i -- python3 bench.py --frames $F --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmc_$i/*/*counter_collection.csv")
if not f: print("no counter file for set $i: $set"); raise SystemExit
tot=collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "k_intra_level" in r["Kernel_Name"]:
        tot[r["Counter_Name"]]+=float(r["Counter_Value"])
print({k:int(v) for k,v in tot.items()})
PY
done
cat gpurun_out/counters_list.txt
