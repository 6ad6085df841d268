# kernel trace of Trainer.training_step in a loop: what the host logic adds on the GPU.  bash tools/r4_trainer_loop_trace.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/tl
rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/bench_trainer_loop.py --batch 32 --steps 12 > $O/log.txt 2>&1
tail -1 $O/log.txt
timeout 60 python3 $R/tools/trainer_loop_trace_summary.py $O < /dev/null
