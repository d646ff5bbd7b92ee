#!/bin/bash
# tools/tiling_matrix.sh [frames] -- the chain on uniform tilings (one transform size per picture) beside the mixed one
cd "$(dirname "$0")/.."
F=${1:-2048}
for t in 4 8 16 32 mix; do
  for dir in "" "--decode"; do
    timeout -k 10 300 python3 bench.py --frames $F --tiling $t $dir --no-fresh --no-ra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); rf=r['roofline']
print('tiling $t $dir F=%d %.0f Mpx/s %.2f ms frac %.4f' % (r['config']['pictures_per_gpu'], r['value'], r['ms_per_step'], rf['frac']))" || exit 1
  done
done
