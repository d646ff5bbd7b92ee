#!/bin/bash
# secondary (inter) workloads of bench.py
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], "Mpx/s", d["ms_per_step"], "ms/step")'
timeout -k 10 600 python bench.py --workload ra1080p8 --segments 64 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ra1080p8 64 segments"
timeout -k 10 600 python bench.py --workload ra1080p8 --segments 128 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ra1080p8 128 segments"
timeout -k 10 600 python bench.py --workload ra2160p8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ra2160p8 default (16 segments)"
timeout -k 10 600 python bench.py --workload ra2160p8 --segments 32 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ra2160p8 32 segments"
timeout -k 10 600 python bench.py --workload ldp1080p8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ldp1080p8 default (16 sequences)"
timeout -k 10 600 python bench.py --workload ldp1080p8 --segments 64 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ldp1080p8 64 sequences"
