# it/s + per-kernel times of library variants with dc_valu=4: tools/exp_dca2.sh base w128 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in "$@"; do
  O=$R/gpurun_out/exp_$f; rm -rf $O; mkdir -p $O
  if [ $f = base ]; then L=$R/helmnet_amd/libhelmnet_hip.so; else L=$R/tools/lib_$f.so; fi
  echo "== $f"
  for rep in 1 2; do timeout 300 python3 $R/tools/run_with_lib.py $L --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt dc_valu=4 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/run_with_lib.py $L --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --opt dc_valu=4 > /dev/null 2>&1
  python3 $R/tools/kstats.py $O k_
done
