#!/bin/bash
# the merged level-0 launch in every mode: parity tests, which kernels run (rocprof), and the headline beside the r5 form of the launch
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/r6pm; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "merged_level0 or one_launch or lanes_replay or capture or graph" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for m in eager graph lanes; do
  rm -rf $O/kt_$m; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -- python3 tools/pair_modes_check.py $m > $O/kt_$m.log 2>&1
  tail -1 $O/kt_$m.log
  python3 tools/kstats.py $O/kt_$m | grep -E "k_dc_asm|k_deepx|k_dc_mfma_s<8, 2" | cut -c1-150
  find $O/kt_$m -name "*kernel_trace.csv" -delete; find $O/kt_$m -name "*agent_info.csv" -delete
done
for i in 1 2; do python bench.py --gpus 1 --steps 300 --warmup 20 > $O/bench_$i.json 2> $O/bench_$i.err; python3 -c "
import json,sys; d=json.load(open('$O/bench_$i.json')); print('bench', d['value'], d['roofline'].get('dominant', d['roofline'].get('kernel')), d['roofline']['frac'])"; done
