#!/bin/bash
# average k_intra_level duration per uniform tiling (level schedule, 64 pictures 1080p)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 4 8 16 32 mix; do
  rm -rf gpurun_out/pt_$t
  HMX_INTRA_SCHEDULE=level timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pt_$t -- python3 bench.py --workload ai1080p8 --frames 64 --steps 1 --warmup 1 --no-cpu-baseline --tiling $t > /dev/null 2>&1
  echo "tiling $t: $(grep k_intra_level gpurun_out/pt_$t/*/*kernel_stats.csv | awk -F, '{printf "calls %s avg_us %.2f min %.2f max %.2f", $2, $4/1000, $6/1000, $7/1000}')"
done
