cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in "$@"; do
  O=$R/gpurun_out/exp_$f; rm -rf $O; mkdir -p $O
  if [ $f = base ]; then L=$R/helmnet_amd/libhelmnet_hip.so; else L=$R/tools/lib_$f.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/run_with_lib.py $L --steps 30 --warmup 10 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  echo "== $f"; python3 $R/tools/kstats.py $O k_dc_wino
done
