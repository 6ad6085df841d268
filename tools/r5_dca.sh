#!/bin/bash
# r5: hand-scheduled level-0 DoubleConv (hn_dca.hip, HN_OPT_DC_VALU 3 / 4) vs hn_dcv.hip (1 / 2): parity, interleaved runs, per-kernel times
cd "$(dirname "$0")/.." && O=gpurun_out/${1:-r5b} && mkdir -p $O && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "vector_fma" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  for m in 1 4 6 5; do
    echo "== dc_valu=$m rep $rep" >> $O/ab256.txt
    timeout 300 $B --opt dc_valu=$m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], [(k['kernel'],k['us']) for k in d['level0_kernels']])" >> $O/ab256.txt
  done
done
for m in 1 4 6; do
  echo "== 512 dc_valu=$m" >> $O/ab512.txt
  timeout 300 $B --size 512 --batch 16 --steps 150 --opt dc_valu=$m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])" >> $O/ab512.txt
done
cat $O/ab256.txt $O/ab512.txt
for m in 6; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$m -- python3 bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary --opt dc_valu=$m > $O/prof$m.log 2>&1
  python tools/kstats.py $O/prof$m k_ > $O/kstats$m.txt; cat $O/kstats$m.txt
done
