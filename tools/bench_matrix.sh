#!/bin/bash
# bench.py over batch sizes / plan counts / schedules (one JSON line each) -> gpurun_out/<tag>_bench_matrix.txt
# usage: tools/bench_matrix.sh TAG frames:plans:schedule[:extra bench args] ...
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}; shift
OUT=gpurun_out/${TAG}_bench_matrix.txt
: > $OUT
run() { echo "## $*" >> $OUT; timeout -k 10 400 "$@" >> $OUT 2>gpurun_out/${TAG}_bm.err || { echo "FAILED rc=$?" >> $OUT; tail -5 gpurun_out/${TAG}_bm.err >> $OUT; }; }
for spec in "$@"; do
  IFS=: read F P S X <<< "$spec"
  if [ "$S" = "packed" ] || [ -z "$S" ]; then run python3 bench.py --frames $F --plans $P --no-cpu-baseline --no-ra $X
  else echo "## HMX_INTRA_SCHEDULE=$S" >> $OUT; HMX_INTRA_SCHEDULE=$S run python3 bench.py --frames $F --plans $P --no-cpu-baseline --no-ra $X; fi
done
grep -E '^(##|\{|FAILED)' $OUT | python3 -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): print(l); continue
    d=json.loads(l); r=d['roofline']
    print('   value %.1f Gpx/s  step %.2f ms  chain frac %.3f  step frac %.3f  conv %s  kernel %s' % (d['value']/1e3, d['ms_per_step'], r['frac'], r['frac_step'], r['layout_conversion_ms'], r['kernel']))
"
