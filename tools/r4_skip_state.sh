# timing only: which level's hidden-state DoubleConv (side stream) costs the iteration what?  HN_EXP_SKIP_STATE bit d skips conv_state_d (results are WRONG)
# build first (here): bash tools/build_variant.sh skipstate hn_unet.hip -DHN_EXP_SKIP_STATE
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for m in 0 1 2 4 6 7; do
  echo -n "skip mask $m: "; HN_EXP_SKIP_STATE=$m timeout 200 python3 $R/tools/run_with_lib.py $R/tools/lib_skipstate.so --steps 300 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
