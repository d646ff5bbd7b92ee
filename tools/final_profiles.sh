#!/bin/bash
# Round profiles of the default bench: kernel-trace stats, the bench line under the profiler, the plain
# bench line, and the PMC traffic passes.  Copies the summaries to gpurun_out/final/ (then into profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r01}; F=${2:-1792}
mkdir -p gpurun_out/final
rm -rf gpurun_out/fp_stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fp_stats -- python3 bench.py --frames $F --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/${TAG}_bench_under_rocprof.json 2> gpurun_out/fp_stats.err || exit 1
cp gpurun_out/fp_stats/*/*kernel_stats.csv gpurun_out/final/${TAG}_kernel_stats.csv
bash tools/collect_traffic.sh $F $TAG || exit 1
cp gpurun_out/${TAG}_traffic.json gpurun_out/final/
cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_traffic.json   # bench.py reads it for roofline.traffic
timeout -k 10 600 python3 bench.py --frames $F > gpurun_out/final/${TAG}_bench_default.json 2> gpurun_out/fp_bench.err || exit 1
tail -1 gpurun_out/final/${TAG}_bench_default.json
head -4 gpurun_out/final/${TAG}_kernel_stats.csv
