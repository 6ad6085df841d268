#!/bin/bash
# upper bound of taking conv_state_0 off the side stream: HN_EXP_SKIP_STATE bit e skips conv_state_e (results WRONG; timing only); x k_deepx with 8 / 16 wavefronts
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in lib_skipstate lib_skipstate1024; do
for v in 0 1 3; do
  export HELMNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/$lib.so HN_EXP_SKIP_STATE=$v
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32 $lib skip=$v', d['value'])"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16 $lib skip=$v', d['value'])"
done; done; done
