#!/bin/bash
# the 16-bit modes with the deep levels as one launch (k_deepx, fp32 arithmetic as before) and the flag sync of the side stream: options flipped, interleaved
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "fp16 or bf16 or precision or config5 or mixed" 2>&1 | tail -3
for rep in 1 2; do
for prec in fp16 bf16x3; do
for o in "deep=2 side_sync=1" "deep=1 side_sync=1" "deep=1 side_sync=0"; do
  set -- $o
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --precision $prec --opt $1 --opt $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('256x32 $prec $o', d['value'], d['residual_rmse_after_timed_steps']['median'])"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --size 512 --batch 16 --precision $prec --opt $1 --opt $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512x16 $prec $o', d['value'])"
done; done; done
