# A/B of the Winograd level-0 DoubleConvs on one box: interleaved 300-step runs + per-kernel times from rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
for rep in 1 2; do
  for m in 11 0 3 8; do
    python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --opt dc_wino=$m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dc_wino=$m', 'rep $rep', round(d['value'],1), 'it/s', d['ms_per_step'], 'ms', d.get('roofline',{}).get('achieved'))"
  done
done > $O/ab_bench.txt 2>&1
for m in 11 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_wino$m -- python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary --opt dc_wino=$m > $O/kt_wino$m.log 2>&1
  echo "== dc_wino=$m" >> $O/ab_bench.txt
  python3 $R/tools/kstats.py $O/kt_wino$m | sort -k5 -n -r | head -24 >> $O/ab_bench.txt
done
cat $O/ab_bench.txt
