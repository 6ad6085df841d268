# two pipeline lanes (the halves of the batch as two chains) on today's kernels: bash tools/r4_lanes.sh
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
 for cfg in "--lanes 1" "--lanes 2" "--lanes 2 --opt side_stream=0" "--lanes 1 --opt side_stream=0" "--lanes 2 --batch 64" "--lanes 1 --batch 64"; do
  echo -n "$cfg: "; timeout 200 python3 $R/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary $cfg 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
 done
done
