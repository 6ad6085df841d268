#!/bin/bash
cd "$(dirname "$0")/.." && O=gpurun_out/r5g && mkdir -p $O && export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "one_launch or vector_fma" 2>&1 | tail -4
for rep in 1 2 3; do for pr in 1 0; do
  echo -n "dc_pair $pr rep $rep: " | tee -a $O/ab.txt
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt dc_pair=$pr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['runner_up']['kernel'], r['runner_up']['avg_launch_us'])" | tee -a $O/ab.txt
done; done
for pr in 1 0; do echo -n "512 dc_pair $pr: " | tee -a $O/ab.txt; timeout 300 python bench.py --size 512 --batch 16 --steps 150 --warmup 30 --no-cpu-baseline --no-secondary --opt dc_pair=$pr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/kstats.py $O/prof k_dc_asm
