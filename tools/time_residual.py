"""Diagnostic: the spectral residual timed alone (back-to-back launches) versus inside the solver loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0"); s.set_domain_size(256, source_location=[30, 128])
B = 32
sos = torch.from_numpy(ring_sos_batch(256, B, seed=0)).cuda()
eng = s.engine()
k_sq, wf = s.get_initials(sos)
src = s._src()
wf = torch.randn_like(wf) * 1e-3
for mode in ("alone", "alone-after-heavy"):
    if mode == "alone-after-heavy":
        out = s.forward(sos, num_iterations=300, residuals="norms")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        r = eng.residual(wf, k_sq, src)
    e1.record(); torch.cuda.synchronize()
    print(mode, "residual (cols+rows) us per call:", e0.elapsed_time(e1) / 200 * 1e3)
