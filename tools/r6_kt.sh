# per-kernel rocprofv3 durations of the bench loop: gpurun -- bash tools/r6_kt.sh <tag> [bench args]   -> gpurun_out/<tag>/kstats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/$tag
mkdir -p $O; rm -rf $O/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary "$@" > $O/kt.log 2>&1
python3 $R/tools/kstats.py $O/kt | sort -k5 -n -r | head -26 > $O/kstats.txt
find $O/kt -name "*kernel_trace.csv" -delete; find $O/kt -name "*agent_info.csv" -delete
cat $O/kstats.txt
