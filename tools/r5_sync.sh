#!/bin/bash
# A/B of HN_OPT_SIDE_SYNC on one box: parity first, then interleaved headline runs, the 512 config, other batches.
cd "$(dirname "$0")/.." && O=gpurun_out/r5s && mkdir -p $O && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "device_flags or one_launch or side_stream" 2>&1 | tail -4 | tee $O/pytest.txt
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for rep in 1 2 3; do for v in 1 0; do
  echo -n "256x32 side_sync $v rep $rep: " | tee -a $O/ab.txt
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt side_sync=$v 2>/dev/null | python -c "$P" | tee -a $O/ab.txt
done; done
for rep in 1 2; do for v in 1 0; do
  echo -n "512x16 side_sync $v rep $rep: " | tee -a $O/ab.txt
  timeout 300 python bench.py --size 512 --batch 16 --steps 150 --warmup 30 --no-cpu-baseline --no-secondary --opt side_sync=$v 2>/dev/null | python -c "$P" | tee -a $O/ab.txt
done; done
for bt in 8 64; do for v in 1 0; do
  echo -n "256x$bt side_sync $v: " | tee -a $O/ab.txt
  timeout 300 python bench.py --batch $bt --steps 200 --warmup 30 --no-cpu-baseline --no-secondary --opt side_sync=$v 2>/dev/null | python -c "$P" | tee -a $O/ab.txt
done; done
for v in 1 0; do echo -n "default --steps 20 side_sync $v: " | tee -a $O/ab.txt; timeout 300 python bench.py --no-cpu-baseline --no-secondary --opt side_sync=$v 2>/dev/null | python -c "$P" | tee -a $O/ab.txt; done
