#!/bin/bash
# tools/ab.sh "libA libB ..." [bench args] -- the default bench (headline only) on several builds of the library, one line each
cd "$(dirname "$0")/.."
LIBS=$1; shift
for l in $LIBS; do
  p=thevc_amd/$l
  echo "== $l $*"
  HMX_LIB_PATH=$PWD/$p timeout -k 10 300 python3 bench.py --steps 4 --warmup 1 --no-ra --no-cpu-baseline --no-fresh "$@" 2>&1 | grep -o '"value": [0-9.]*, "unit": "Mpixels/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' | head -1
done
