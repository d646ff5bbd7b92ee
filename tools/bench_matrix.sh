#!/bin/bash
# the table of DESIGN.md section 5: batch sizes, uniform tilings, schedules, secondary workloads
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get("roofline") or {}; print(sys.argv[1], d["value"], "Mpx/s", d["ms_per_step"], "ms/step", "frac", r.get("frac"), "launch_us", r.get("avg_launch_us"), "conv_ms", r.get("layout_conversion_ms"))'
for f in 8 64 256 512 1024 1536 1792; do timeout -k 10 600 python bench.py --frames $f --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "mix F=$f"; done
for t in 4 8 16 32; do timeout -k 10 600 python bench.py --frames 1536 --steps 2 --warmup 1 --no-cpu-baseline --tiling $t 2>/dev/null | python -c "$P" "tiling $t F=1536"; done
HMX_INTRA_ACROSS=0 timeout -k 10 600 python bench.py --frames 1536 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "per-picture level kernel F=1536"
HMX_INTRA_SCHEDULE=wave timeout -k 10 600 python bench.py --frames 8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "wave schedule F=8"
timeout -k 10 600 python bench.py --workload ai1080p8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ai1080p8 default"
timeout -k 10 600 python bench.py --workload ai2160p8 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "ai2160p8 default"

timeout -k 10 600 python bench.py --decode --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "decoder direction default"
HMX_PIPELINE_CONV=1 timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P" "pipelined conversions default"
