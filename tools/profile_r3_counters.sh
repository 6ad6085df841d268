# L2 / fabric counters of the default (XCD-aware tile order) and -DHN_NO_XCD builds:
#   gpurun -- 'bash tools/profile_r3_counters.sh'   (tools/lib_noxcd.so must have been built: every source with -DHN_NO_XCD)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/tcc_counters_available.txt
ARGS="--steps 24 --warmup 12 --no-cpu-baseline --no-secondary"
for v in xcd noxcd; do
  if [ $v = xcd ]; then PROG="python3 $R/bench.py $ARGS"; else PROG="python3 $R/tools/run_with_lib.py $R/tools/lib_noxcd.so $ARGS"; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${v}_fetch -- $PROG > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${v}_write -- $PROG > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${v}_hit -- $PROG > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/${v}_rdreq -- $PROG > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/${v}_req -- $PROG > /dev/null 2>&1
done
cd $R
python3 tools/summarize_tcc.py $O > $O/summary.txt; cat $O/summary.txt | head -60
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
