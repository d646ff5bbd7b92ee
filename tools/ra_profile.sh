#!/bin/bash
# Kernel-trace stats of a random-access / low-delay bench run (which kernels the inter pictures spend their time in).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-ra1080p8}; S=${2:-64}; TAG=${3:-ra}
rm -rf gpurun_out/${TAG}_stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --workload $W --segments $S --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_stats.err || exit 1
cp gpurun_out/${TAG}_stats/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/${TAG}_bench.json
head -25 gpurun_out/${TAG}_kernel_stats.csv
