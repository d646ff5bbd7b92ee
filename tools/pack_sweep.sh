#!/bin/bash
# knob sweep of the packed schedule: tools/pack_sweep.sh TAG "frames plans [ENV=V ...]" ...
cd $GRAFT_REPO_ROOT
TAG=${1:-sweep}; shift
OUT=gpurun_out/${TAG}_pack_sweep.txt
: > $OUT
for spec in "$@"; do
  set -- $spec
  F=$1; P=$2; shift 2
  line=$(env "$@" timeout -k 10 300 python3 bench.py --frames $F --plans $P --distinct 4 --steps 3 --warmup 1 --no-cpu-baseline --no-ra $EXTRA 2>gpurun_out/${TAG}_sweep.err | tail -1)
  echo "$F $P $* :: $(echo "$line" | python3 -c "
import sys, json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('%.1f Gpx/s  step %.2f ms  chain %.2f ms  frac %.3f' % (d['value']/1e3, d['ms_per_step'], r['avg_launch_us']*r['launches_per_step']/1e3, r['frac']))
except Exception as e:
    print('FAILED', e)
")" | tee -a $OUT
done
