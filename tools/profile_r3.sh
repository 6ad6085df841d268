# Round-3 evidence: run on the GPU box as  gpurun -- 'bash tools/profile_r3.sh'  (writes gpurun_out/r3p/, condensed by tools/summarize_profiles.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
rm -rf $O; mkdir -p $O
BENCH="python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_wave -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/pmc_inst -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python3 $R/tools/bench_train.py --steps 5 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/bench_k300.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt spectral_cols=0 > $O/bench_k300_cols0.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 > $O/bench_512.json 2>/dev/null
python tools/bench_train.py --cpu > $O/bench_train.json 2>/dev/null
python tools/bench_train.py --batch 128 --steps 5 > $O/bench_train_b128.json 2>/dev/null
python tools/bench_train.py --n 256 --batch 8 --steps 5 > $O/bench_train_256.json 2>/dev/null
python tools/ab_cols.py 2>/dev/null > $O/cols_ab.txt
python tools/ab_512.py 2>/dev/null > $O/ab_512.txt
tools/bin/ubench_launch_floor > $O/ubench_launch_floor.txt 2>/dev/null
tools/bin/ubench_two_queues > $O/ubench_two_queues.txt 2>/dev/null
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
du -sh $O; cat $O/pytest_gpu.txt; cat $O/bench_k20.json | cut -c1-400; cat $O/bench_train*.json
