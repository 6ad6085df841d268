# weight-gradient launches with a capped block count beside the backward chain, other shapes: bash tools/r4_train_cumask.sh
R=$GRAFT_REPO_ROOT
for shape in "256 8" "128 32" "64 32" "96 64" "96 8"; do set -- $shape
 for cap in 0 640 1280; do for o in 0 1; do
   if [ $o = 0 ] && [ $cap != 0 ]; then continue; fi
   echo -n "n=$1 batch=$2 cap=$cap train_overlap=$o: "; HN_EXP_WG_CAP=$cap timeout 120 python3 $R/tools/bench_train.py --steps 10 --n $1 --batch $2 --opt train_overlap=$o 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"
 done; done
done
