# per-kernel times of hn_dca.hip build variants: tools/exp_dca.sh base a4 a5 ...   (variants from tools/build_variant.sh <name> hn_dca.hip -DHN_AEXP=<bits>)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in "$@"; do
  O=$R/gpurun_out/exp_$f; rm -rf $O; mkdir -p $O
  if [ $f = base ]; then L=$R/helmnet_amd/libhelmnet_hip.so; else L=$R/tools/lib_$f.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/run_with_lib.py $L --steps 30 --warmup 10 --no-cpu-baseline --no-secondary --opt dc_valu=4 > /dev/null 2>&1
  echo "== $f"; python3 $R/tools/kstats.py $O k_dc_asm
done
