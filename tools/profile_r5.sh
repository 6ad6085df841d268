# Round-5 evidence: run on the GPU box as  gpurun -- 'bash tools/profile_r5.sh'  (writes gpurun_out/r5p/, condensed by tools/summarize_profiles.py gpurun_out/r5p r5)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5p
rm -rf $O; mkdir -p $O
BENCH="python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_wave -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/pmc_inst -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc_mfma -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_512 -- python3 $R/bench.py --steps 30 --warmup 12 --no-cpu-baseline --no-secondary --size 512 --batch 16 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python3 $R/tools/bench_train.py --steps 5 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/bench_k300.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt dc_valu=1 > $O/bench_k300_dcvalu1.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/bench_k300_b.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 > $O/bench_512.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --train --steps 10 --warmup 3 > $O/bench_train_ddp1.json 2>/dev/null
python tools/bench_train.py > $O/bench_train.json 2>/dev/null
python tools/power_probe.py 4 1 0 > $O/power_probe.txt 2>/dev/null
python tools/gmres_vs_learned.py --budget 20 > $O/gmres_vs_learned.json 2>/dev/null
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -40 > $O/pytest_gpu.txt
du -sh $O; tail -3 $O/pytest_gpu.txt; cut -c1-400 $O/bench_k20.json; cat $O/power_probe.txt
