#!/bin/bash
# device-derived epoch of the merged level-0 launch against the r5 form (epoch in the kernel arguments): the same box, interleaved.
#   tools/_ab_hostepoch.so = the library of the commit before (python -m helmnet_amd.build in a checkout of it); not tracked
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in new old; do
  if [ $v = old ]; then export HELMNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/_ab_hostepoch.so; else unset HELMNET_HIP_LIB; fi
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32 $v', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['runner_up']['avg_launch_us'])"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16 $v', d['value'])"
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt graph=1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32 graph=1 $v', d['value'])"
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --batch 8 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x8 $v', d['value'])"
done; done
