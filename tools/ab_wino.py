"""A/B of the Winograd level-0 DoubleConvs (hn_wino.hip) against the direct kernels, one kind at a time, with error locations.
   python tools/ab_wino.py [n] [batch]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from golden_inputs import teacher_inputs
from helmnet_amd import IterativeSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ti = {k: torch.from_numpy(v) for k, v in teacher_inputs(n, b, seed=4242).items()}
outs = {}
for mask in (0, 1, 2, 8, 11):
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
    s.set_domain_size(n, source_location=[n // 3, n // 2])
    s.engine().set_option("dc_wino", mask)
    g = {k: v.to("cuda:0") for k, v in ti.items()}
    k_sq, _ = s.get_initials(g["sos"])
    s.f.set_states(g["states"], flatten=True)
    wf2, res2 = s.single_step(g["wf"], k_sq, g["res"])
    torch.cuda.synchronize()
    outs[mask] = (wf2.cpu(), res2.cpu(), s.f.get_states(flatten=True).cpu())
for mask in (1, 2, 8, 11):
    for name, a, d in zip(("wf", "res", "states"), outs[mask], outs[0]):
        diff = (a - d).abs()
        scale = d.abs().max().item()
        idx = np.unravel_index(int(diff.argmax()), diff.shape)
        print(f"mask {mask:2d} {name:6s} max|diff| / max = {diff.max().item() / scale:.3e}  at {idx}  nan={bool(torch.isnan(a).any())}")
        if name == "wf" and diff.max().item() / scale > 1e-5:
            bad = (diff[0].amax(0) > 1e-5 * scale).numpy()
            ys, xs = np.nonzero(bad)
            print("   bad pixels:", bad.sum(), "rows", np.unique(ys)[:40], "cols", np.unique(xs)[:40])
