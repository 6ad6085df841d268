"""A/B of the 512-point spectral kernels: HN_OPT_SPECTRAL_RADIX16 0 (radix-4 Stockham) vs 1 (8 x 8 x 8 register-resident, r3): agreement and time of hn_residual alone."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from helmnet_amd import IterativeSolver
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(512, source_location=[450, 256])
eng = s.engine()
src = s.source.detach().contiguous()
for b in (16, 2):
    wf = torch.randn(b, 2, 512, 512, device="cuda:0"); ksq = torch.rand(b, 1, 512, 512, device="cuda:0") + 0.5
    ref = None
    for mode in (0, 1, 0, 1):
        eng.set_option("spectral_radix16", mode)
        out = eng.residual(wf, ksq, src)
        for _ in range(20): eng.residual(wf, ksq, src)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): eng.residual(wf, ksq, src)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        if mode == 0: ref = out
        print(f"B={b} radix16={mode}: {dt * 1e6:.1f} us per residual  max|diff| vs radix-4 / max|res| = {float((out - ref).abs().max() / ref.abs().max()):.2e}")
