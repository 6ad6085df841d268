#!/bin/bash
cd "$(dirname "$0")/.." && O=gpurun_out/r5dbg && mkdir -p $O && export TMPDIR=/tmp
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], [(k["kernel"],k["bracketed_us"]) for k in d["level0_kernels"]]); [print("   ", x["value"], x["unit"], x["workload"][:60]) for x in d.get("secondary", [])]'
echo "== main 512 (default opts)"; python bench.py --size 512 --batch 16 --steps 150 --warmup 30 --no-cpu-baseline --no-secondary --breakdown 2> $O/bd512.txt | python -c "$P"; head -12 $O/bd512.txt
echo "== main 512 dc_valu=1"; python bench.py --size 512 --batch 16 --steps 150 --warmup 30 --no-cpu-baseline --no-secondary --opt dc_valu=1 2>/dev/null | python -c "$P"
echo "== default 256 with secondary"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "== default 256 with secondary, dc_valu=1 via env"; HN_DC_VALU=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$P"
