# where the side stream's hidden-state kernels are released (deep kernel vs the last layer-by-layer down) x deepx with 8 / 16 wavefronts (r6)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "helmnet_amd/libhelmnet_hip.so 0" "helmnet_amd/libhelmnet_hip.so 1" "tools/lib_dx1024.so 0" "tools/lib_dx1024.so 1"; do
  set -- $cfg
  echo "== $1 HN_EXP_REL_DOWN=$2"
  HN_EXP_REL_DOWN=$2 HELMNET_HIP_LIB=$1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32', d['value'])"
  HN_EXP_REL_DOWN=$2 HELMNET_HIP_LIB=$1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16', d['value'])"
done; done
