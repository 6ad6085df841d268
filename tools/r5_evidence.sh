#!/bin/bash
# r5: ablations, block timeline, pk_fma micro-benchmark of hn_dca.hip -> gpurun_out/r5ev/*.txt
cd "$(dirname "$0")/.." && O=gpurun_out/r5ev && mkdir -p $O && export TMPDIR=/tmp
bash tools/exp_dca.sh base a4 a5 a12 a28 a30 a6 a20 > $O/dca_ablation.txt 2>&1
python tools/dca_trace.py tools/lib_trace.so 4 > $O/dca_trace.txt 2>/dev/null
tools/bin/ubench_pk > /dev/null 2>&1; tools/bin/ubench_pk > $O/ubench_pk.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "vector_fma" 2>&1 | tail -3
cat $O/dca_ablation.txt | grep -c "k_dc_asm"; head -12 $O/dca_trace.txt; tail -4 $O/ubench_pk.txt
