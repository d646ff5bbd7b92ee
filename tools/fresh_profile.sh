#!/bin/bash
# tools/fresh_profile.sh [frames] -- per-kernel times of the fresh-decisions leg (plans on the device, tables, chain)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
F=${1:-2048}
O=gpurun_out/fresh_prof_$F
rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --frames $F --steps 1 --warmup 0 --fresh-steps 2 --no-ra --no-cpu-baseline > $O.log 2>&1
tail -c 1200 $O.log
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:16]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:10.3f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
rm -rf $O/*/*kernel_trace.csv
