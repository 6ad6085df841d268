# Round-6 evidence: run on the GPU box as  gpurun -- 'bash tools/profile_r6.sh'  (writes gpurun_out/r6p/ and gpurun_out/r6p512/; condensed by
# tools/summarize_profiles.py gpurun_out/r6p r6  and  tools/summarize_profiles.py gpurun_out/r6p512 r6_512)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p
P=$R/gpurun_out/r6p512
rm -rf $O $P; mkdir -p $O $P
BENCH="python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary"
B512="python3 $R/bench.py --steps 30 --warmup 12 --no-cpu-baseline --no-secondary --size 512 --batch 16"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_wave -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/pmc_inst -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc_mfma -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- $B512 > $P/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- $B512 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- $B512 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $P/pmc_wave -- $B512 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $P/pmc_mfma -- $B512 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -- python3 $R/tools/bench_train.py --steps 5 > /dev/null 2>&1
find $O $P -name "*kernel_trace.csv" -delete; find $O $P -name "*agent_info.csv" -delete
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/bench_k300.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt deep=1 --opt inc_sigma_map=0 > $O/bench_k300_r5kernels.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/bench_k300_b.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 > $O/bench_512.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 --opt deep=1 --opt inc_sigma_map=0 > $O/bench_512_r5kernels.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --train --steps 10 --warmup 3 > $O/bench_train_ddp1.json 2>/dev/null
python tools/bench_train.py > $O/bench_train.json 2>/dev/null
python tools/deepx_trace.py tools/lib_dxtrace.so 256 32 > $O/deepx_trace.txt 2>/dev/null
python tools/deepx_trace.py tools/lib_dxtrace.so 512 16 >> $O/deepx_trace.txt 2>/dev/null
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "Warning\|warnings.warn\|^$\|amdgpu.ids" | tail -60 > $O/pytest_gpu.txt
du -sh $O $P; tail -3 $O/pytest_gpu.txt; cut -c1-300 $O/bench_k20.json
