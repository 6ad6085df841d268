# HN_OPT_TRAIN_OVERLAP modes across shapes: bash tools/r4_train_modes.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for shape in "96 32" "96 8" "64 32" "128 32" "256 8" "96 128"; do set -- $shape
 for o in 0 2; do
   echo -n "n=$1 batch=$2 train_overlap=$o: "; timeout 120 python3 $R/tools/bench_train.py --steps 10 --n $1 --batch $2 --opt train_overlap=$o 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"
 done
done
done
