#!/bin/bash
# per-kernel durations of tools/inter_bench.py (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WHAT=${1:-all}; N=${2:-64}
rm -rf gpurun_out/is_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/is_stats -- python3 tools/inter_bench.py $N $WHAT > gpurun_out/is_bench.log 2> gpurun_out/is_stats.err || exit 1
cat gpurun_out/is_bench.log
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/is_stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY
