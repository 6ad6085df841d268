"""Soak test of the device-side hand-offs in the modes that derive their epochs on the device (r6): captured iterations replayed (HN_OPT_GRAPH), two pipeline
lanes, the 16-bit modes with k_deepx -- thousands of iterations each, every device-side wait checked (hn_check_async_errors), results finite, converged residual
equal to the plain fp32 loop's to the network's noise floor.
    python tools/modes_soak.py [iterations]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
dev = "cuda:0"
n, b = 256, 32
sos = torch.from_numpy(ring_sos_batch(n, b, seed=7)).to(dev)
for name, prec, opts in (("plain", "fp32", {}), ("graph", "fp32", {"graph": 1}), ("lanes=2", "fp32", {"lanes": 2}), ("fp16", "fp16", {}), ("bf16x3", "bf16x3", {}),
                         ("fp16 + graph", "fp16", {"graph": 1})):
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to(dev); s.set_unet_precision(prec)
    s.set_domain_size(n, source_location=[30, 128])
    eng = s.engine()
    for k, v in opts.items():
        eng.set_option(k, v)
    t0 = time.perf_counter()
    done, o = 0, None
    while done < iters:
        chunk = min(2000, iters - done)
        o = s.forward(sos, num_iterations=chunk, residuals="norms") if done == 0 else s.n_steps(o["wavefields"][0], s.get_initials(sos)[0].contiguous(), o["last_residual"], chunk, residuals="norms")
        done += chunk
    torch.cuda.synchronize()
    eng.check_async_errors()
    dt = time.perf_counter() - t0
    rm = o["residual_norms"][-1]
    assert torch.isfinite(o["wavefields"][0]).all() and torch.isfinite(rm).all()
    print(f"{name:14s} {iters} iterations at {iters / dt:6.0f} it/s, RMSE median {rm.median().item():.3e} max {rm.max().item():.3e}; no device-side wait gave up")
