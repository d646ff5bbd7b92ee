cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --frames 64 --workload ai1080p8 --steps 6 --warmup 2 --no-ra --no-cpu-baseline --no-fresh 2>&1 | grep -o '"value": [0-9.]*, "unit": "Mpixels/s", "n_gpus": 1, "steps": [0-9]*, "warmup": [0-9]*, "ms_per_step": [0-9.]*' | head -1; }
run A=1
run HMX_PACK_SLOTS4=16
run HMX_PACK_GROUP=2
run HMX_PACK_GROUP=4
run HMX_PACK_GROUP=4 HMX_PACK_SLOTS4=16
run HMX_PACK_SLEEP0=0 HMX_PACK_SLEEP1=0
run HMX_PACK_SLEEP0=1 HMX_PACK_SLEEP1=1
run HMX_PACK_WAVES=1024
run HMX_PACK_WAVES=8192
