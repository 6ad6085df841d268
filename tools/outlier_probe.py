"""Are short runs sporadically slow?  Repeats small timed runs of one configuration and prints every figure: python tools/outlier_probe.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from helmnet_amd import IterativeSolver
dev = torch.device("cuda", 0)
solver = IterativeSolver.from_exported_weights(); solver.freeze(); solver.to(dev)
for dc in (4, 1, 4, 1):
    for (n, B, steps, warm) in ((256, 32, 100, 20), (512, 16, 40, 10)):
        vals = []
        for rep in range(8):
            solver.set_unet_precision("fp32")
            eng, _, (wf, res, st, k_sq, src) = bench.make_problem(solver, n, B, [n - 62, n // 2], 5, dev, False)
            eng.set_option("dc_valu", dc)
            eng.step(wf, res, st, k_sq, src, warm); torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.step(wf, res, st, k_sq, src, steps); torch.cuda.synchronize()
            vals.append(steps / (time.perf_counter() - t0))
        print(f"dc_valu={dc} {n}^2 x {B}, {steps} steps x 8 (fresh set_domain each): " + " ".join(f"{v:.0f}" for v in vals), flush=True)
# the same without re-making the problem (no set_domain / reserve between runs)
for dc in (4, 1):
    eng, _, (wf, res, st, k_sq, src) = bench.make_problem(solver, 512, 16, [450, 256], 5, dev, False)
    eng.set_option("dc_valu", dc)
    vals = []
    for rep in range(12):
        eng.step(wf, res, st, k_sq, src, 10); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.step(wf, res, st, k_sq, src, 40); torch.cuda.synchronize()
        vals.append(40 / (time.perf_counter() - t0))
    print(f"dc_valu={dc} 512^2 x 16, 40 steps x 12 (same problem): " + " ".join(f"{v:.0f}" for v in vals), flush=True)
