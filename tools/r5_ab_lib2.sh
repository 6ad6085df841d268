#!/bin/bash
# interleaved A/B of library builds + selected kernel times: tools/r5_ab_lib2.sh <out> <kernel filter> base <variant> ...
cd "$(dirname "$0")/.." && O=gpurun_out/$1 && mkdir -p $O && export TMPDIR=/tmp; K=$2; shift 2
for rep in 1 2 3; do for f in "$@"; do
  if [ $f = base ]; then L=helmnet_amd/libhelmnet_hip.so; else L=tools/lib_$f.so; fi
  echo -n "$f rep $rep: " | tee -a $O/ab.txt
  timeout 300 python tools/run_with_lib.py $L --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done
for f in "$@"; do
  if [ $f = base ]; then L=helmnet_amd/libhelmnet_hip.so; else L=tools/lib_$f.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$f -- python3 tools/run_with_lib.py $L --steps 40 --warmup 12 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  echo "== $f"; python tools/kstats.py $O/prof_$f $K
done
timeout 300 python tools/run_with_lib.py tools/lib_$3.so --steps 30 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rmse', d['residual_rmse_after_timed_steps'])"
