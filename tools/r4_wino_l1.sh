cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
python3 - <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np, torch
from golden_inputs import teacher_inputs
from helmnet_amd import IterativeSolver
n, b = 256, 2
ti = {k: torch.from_numpy(v) for k, v in teacher_inputs(n, b, seed=4242).items()}
outs = {}
for mask in (0, 48, 59):
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
    s.set_domain_size(n, source_location=[n // 3, n // 2])
    s.engine().set_option("dc_wino", mask)
    g = {k: v.to("cuda:0") for k, v in ti.items()}
    k_sq, _ = s.get_initials(g["sos"])
    s.f.set_states(g["states"], flatten=True)
    wf2, res2 = s.single_step(g["wf"], k_sq, g["res"])
    torch.cuda.synchronize()
    outs[mask] = (wf2.cpu(), res2.cpu(), s.f.get_states(flatten=True).cpu())
for mask in (48, 59):
    print("mask", mask, [float((a - d).abs().max() / d.abs().max()) for a, d in zip(outs[mask], outs[0])])
PY
for rep in 1 2; do
  for m in 0 48 16 32 59 11; do
    python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-secondary --opt dc_wino=$m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dc_wino=$m', 'rep $rep', round(d['value'],1), 'it/s', d['ms_per_step'], 'ms')"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_l1 -- python3 $R/bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-secondary --opt dc_wino=48 > /dev/null 2>&1
python3 $R/tools/kstats.py $O/kt_l1 k_dc_ | sort -k5 -n -r
