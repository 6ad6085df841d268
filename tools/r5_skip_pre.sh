#!/bin/bash
# r5: HN_OPT_SKIP_PRE A/B on one box: parity test, interleaved 300-step runs, per-kernel times
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5a && export TMPDIR=/tmp
O=gpurun_out/r5a
python -m pytest tests/test_gpu_parity.py -q -x -k "skip_half or vector_fma" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  for m in 0 1 5 3; do
    echo "== skip_pre=$m rep $rep" >> $O/ab256.txt
    $B --opt skip_pre=$m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['hbm_path']['us_per_step'])" >> $O/ab256.txt
  done
done
for rep in 1 2; do
  for m in 0 1 3 7; do
    echo "== 512 skip_pre=$m rep $rep" >> $O/ab512.txt
    $B --size 512 --batch 16 --steps 150 --opt skip_pre=$m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])" >> $O/ab512.txt
  done
done
cat $O/ab256.txt $O/ab512.txt
for m in 0 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$m -- python3 bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary --opt skip_pre=$m > $O/prof$m.log 2>&1
  python tools/kstats.py $O/prof$m > $O/kstats$m.txt; cat $O/kstats$m.txt
done
