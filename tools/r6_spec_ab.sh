# spectral row-pass A/B (r6): residual alone + the loop, builds interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in helmnet_amd/libhelmnet_hip.so tools/lib_sphead.so tools/lib_sp512e.so tools/lib_sp256late.so; do
  echo "== $lib"
  HELMNET_HIP_LIB=$lib python tools/time_residual.py 256 32
  HELMNET_HIP_LIB=$lib python tools/time_residual.py 512 16
  HELMNET_HIP_LIB=$lib python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32', d['value'], 'spec', d['hbm_path']['us_per_step'])"
  HELMNET_HIP_LIB=$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16', d['value'], 'spec', d['hbm_path']['us_per_step'])"
done; done
