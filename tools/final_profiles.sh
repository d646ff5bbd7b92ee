#!/bin/bash
# Round profiles of the default bench: the plain bench line, kernel-trace stats of the same command, the PMC passes
# (tools/pmc.py: instruction mix, wait / VALU-busy shares, MFMA, LDS, HBM traffic).  Summaries -> gpurun_out/final/
# (copy them into profiles/).   usage: tools/final_profiles.sh TAG [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02}; shift
mkdir -p gpurun_out/final
timeout -k 10 900 python3 bench.py "$@" > gpurun_out/final/${TAG}_bench_default.json 2> gpurun_out/fp_bench.err || { tail -5 gpurun_out/fp_bench.err; exit 1; }
tail -1 gpurun_out/final/${TAG}_bench_default.json | cut -c1-600
rm -rf gpurun_out/fp_stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fp_stats -- python3 bench.py "$@" --no-cpu-baseline --no-ra --no-fresh > gpurun_out/final/${TAG}_bench_under_rocprof.json 2> gpurun_out/fp_stats.err || { tail -5 gpurun_out/fp_stats.err; exit 1; }
cp gpurun_out/fp_stats/*/*kernel_stats.csv gpurun_out/final/${TAG}_kernel_stats.csv
head -5 gpurun_out/final/${TAG}_kernel_stats.csv
F=$(python3 -c "import json;print(json.loads(open('gpurun_out/final/${TAG}_bench_default.json').read().strip().splitlines()[-1])['config']['pictures_per_gpu'])")
P=$(python3 -c "import json;print(json.loads(open('gpurun_out/final/${TAG}_bench_default.json').read().strip().splitlines()[-1])['config']['distinct_plans'])")
python3 tools/pmc.py --tag ${TAG} --kernel k_intra_packed --timeout 500 --meta "{\"frames\": $F, \"plans\": $P}" -- python3 bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-ra --no-fresh | tail -3
rm -rf gpurun_out/pmc_${TAG}_* gpurun_out/fp_stats
cp gpurun_out/${TAG}_pmc_summary.json gpurun_out/final/
