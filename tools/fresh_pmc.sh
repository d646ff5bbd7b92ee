#!/bin/bash
# tools/fresh_pmc.sh [frames] -- instruction and wait counters of the device plan builder's kernels (fresh-decisions leg)
cd "$(dirname "$0")/.."
F=${1:-512}
for k in k_plan_gather k_pack_fill k_plan_scatter; do
  python3 tools/pmc.py --tag fresh_$k --kernel $k --sets inst,wait,lds --timeout 300 -- python3 bench.py --frames $F --steps 1 --warmup 0 --fresh-steps 1 --no-ra --no-cpu-baseline 2>&1 | grep -E "^\{|failed" | cut -c1-700
done
rm -rf gpurun_out/pmc_fresh_*  # (gpurun merges at most 64 MiB back)
