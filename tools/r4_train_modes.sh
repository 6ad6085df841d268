# HN_OPT_TRAIN_FUSED with / without bit 4 (matrix-core backward DoubleConvs) across shapes: bash tools/r4_train_modes.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for shape in "96 32" "96 8" "64 32" "128 32" "256 8" "96 128"; do set -- $shape
 for o in 23 55; do
   echo -n "n=$1 batch=$2 train_fused=$o: "; timeout 120 python3 $R/tools/bench_train.py --steps 10 --n $1 --batch $2 --opt train_fused=$o 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"
 done
done
done
