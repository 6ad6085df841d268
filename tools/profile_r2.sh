# Round-2 evidence: run on the GPU box as  gpurun -- 'bash tools/profile_r2.sh'  (writes gpurun_out/r2/, condensed by tools/summarize_profiles.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2
rm -rf $O; mkdir -p $O
BENCH="python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $BENCH > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_wave -- $BENCH > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/pmc_inst -- $BENCH > /dev/null 2>&1
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > $O/bench_k300.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --graph 1 > $O/bench_k300_graph1.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt deep=0 > $O/bench_k300_nodeep.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt dc_valu=0 > $O/bench_k300_mfma.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --opt dc_valu=2 > $O/bench_k300_allvalu.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 --opt dc_valu=0 > $O/bench_512_mfma.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 > $O/bench_512.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --precision bf16x3 > $O/bench_bf16x3.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --precision fp16 > $O/bench_fp16.json 2>/dev/null
python tools/graph_ab.py > $O/graph_ab.txt 2>/dev/null
for a in "96 32" "96 32 --dense" "160 32" "160 32 --dense" "320 16" "320 16 --dense" "256 32"; do python tools/time_residual.py $a 2>/dev/null; done > $O/residual_times.txt
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
du -sh $O; cat $O/pytest_gpu.txt
