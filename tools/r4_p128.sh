# level 1 (128^2 at the 256^2 workload) on the all-channels-at-once tile kernel instead of the strip kernel: bash tools/r4_p128.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do for L in $R/helmnet_amd/libhelmnet_hip.so $R/tools/lib_p128.so; do
  echo -n "$(basename $L): "; timeout 200 python3 $R/tools/run_with_lib.py $L --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
echo "512^2 x 16:"
for rep in 1 2; do for L in $R/helmnet_amd/libhelmnet_hip.so $R/tools/lib_p128.so; do
  echo -n "$(basename $L): "; timeout 200 python3 $R/tools/run_with_lib.py $L --size 512 --batch 16 --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
