# quick A/B on one box: bash tools/r4_ab_quick.sh "<opt a>" "<opt b>" ... ; each 300-step run twice + per-kernel stats of the first variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
python3 $R/tools/ab_wino.py 256 2 2>&1 | grep "mask 11"
for rep in 1 2; do
  for m in "$@"; do
    python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --opt $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', 'rep $rep', round(d['value'],1), 'it/s', d['ms_per_step'], 'ms')"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_q -- python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary --opt $1 > $O/kt_q.log 2>&1
python3 $R/tools/kstats.py $O/kt_q k_dc_ k_down_mfma\<4 k_up_mfma\<2 | sort -k5 -n -r
