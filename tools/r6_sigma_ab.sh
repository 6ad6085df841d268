# HN_OPT_INC_SIGMA_MAP A/B (r6): bench loop 300 steps, interleaved, two repetitions + per-kernel times with the map
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in 1 0; do
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt inc_sigma_map=$v | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32 inc_sigma_map=$v', d['value'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 --opt inc_sigma_map=$v | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16 inc_sigma_map=$v', d['value'])"
done; done
