# kernel trace of a few training steps (timeline analysis): bash tools/r4_train_trace.sh [bench_train args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/tt
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_train.py --steps 4 "$@" > $O/log.txt 2>&1
python3 $R/tools/train_trace_summary.py $O
