#!/bin/bash
# same-box A/B of two builds of the library: tools/_ab_old.so (not tracked: a build of the commit before) against the tree's; interleaved, 3 repetitions
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export HELMNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/_ab_old.so; else unset HELMNET_HIP_LIB; fi
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('256x32 $v', d['value'], r['avg_launch_us'], r['runner_up']['avg_launch_us'], 'spectral', d['hbm_path']['us_per_step'])"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16 $v', d['value'], 'spectral', d['hbm_path']['us_per_step'])"
done; done
