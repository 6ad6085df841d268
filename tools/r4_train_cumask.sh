R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for c in "640 160 320" "640 320 320" "640 320 640" "512 256 256" "512 512 512" "768 384 384" "384 192 384" "448 224 448"; do set -- $c
   echo -n "caps $1 $2 $3: "; HN_EXP_CAP8=$1 HN_EXP_CAP2=$2 HN_EXP_CAPK=$3 timeout 120 python3 $R/tools/bench_train.py --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"
done; done
