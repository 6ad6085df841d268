R=$GRAFT_REPO_ROOT
for rep in 1 2; do for shape in "96 32" "96 8" "64 32" "128 32" "256 8" "96 128"; do set -- $shape; for e in 0 1; do
  if [ $e = 0 ]; then export HN_EXP_NO_SIDE_STATE=1; else unset HN_EXP_NO_SIDE_STATE; fi
  echo -n "n=$1 batch=$2 side_state_fwd=$e: "; timeout 120 python3 $R/tools/bench_train.py --steps 10 --n $1 --batch $2 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"
done; done; done
