#!/bin/bash
# where conv_state is released x k_deepx with 8 / 16 wavefronts: HN_EXP_REL_UP=1 = by the first decoder `up` behind the deep kernel (the deep kernel then runs alone)
#   tools/lib_relup.so: hn_unet.hip built with -DHN_EXP_REL_UP; tools/lib_relup1024.so: + hn_deepx.hip with -DHN_DX_NT=1024  (not tracked)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in lib_relup lib_relup1024; do
for v in 0 1; do
  export HELMNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/$lib.so HN_EXP_REL_UP=$v
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32 $lib rel_up=$v', d['value'])"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16 $lib rel_up=$v', d['value'])"
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --batch 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x8 $lib rel_up=$v', d['value'])"
done; done; done
export HELMNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/lib_relup1024.so HN_EXP_REL_UP=1
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "side_stream or deep or lanes_replay" 2>&1 | tail -2
