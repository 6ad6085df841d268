#!/bin/bash
# r5: full GPU test suite + the driver's bench command + default bench + 512 on one box
cd "$(dirname "$0")/.." && O=gpurun_out/${1:-r5full} && mkdir -p $O && export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err; python -c "
import json; d=json.loads(open('$O/bench_k20.json').read().strip().splitlines()[-1]); r=d['roofline']
print('k20', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'], r.get('executed'), 'runner_up', r['runner_up']['kernel'], r['runner_up']['avg_launch_us'], r['runner_up']['frac'], 'tie', r['tie'], 'step', r['step']['frac'], 'hbm_path', d['hbm_path']['us_per_step'])
for x in d.get('secondary', []): print('   ', x['value'], x['unit'], x['workload'][:70])"
