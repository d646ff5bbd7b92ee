#!/bin/bash
# tools/quads_ab.sh -- the chain with and without quads (HMX_PLAN_QUADS) over batch sizes; one line each
cd "$(dirname "$0")/.."
run() { echo "== quads=$1 $2"; HMX_PLAN_QUADS=$1 timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-ra --no-cpu-baseline --no-fresh $2 2>&1 | grep -o '"value": [0-9.]*, "unit": "Mpixels/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*\|[0-9]*-[0-9]* dependency levels' | tr '\n' ' '; echo; }
for a in "" "--frames 256" "--frames 64" "--frames 8" "--workload ai1080p8 --frames 64" "--decode"; do
  run 0 "$a"; run 1 "$a"
done
