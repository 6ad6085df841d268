"""A/B of the 256-point column-pass kernels (HN_OPT_SPECTRAL_COLS 0 / 1 / 2): bit-equality of the residual and time of hn_residual alone."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from helmnet_amd import IterativeSolver
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(256, source_location=[30, 128])
eng = s.engine()
src = s.source.detach().contiguous()
ref = None
for b in (32, 4):
    wf = torch.randn(b, 2, 256, 256, device="cuda:0"); ksq = torch.rand(b, 1, 256, 256, device="cuda:0") + 0.5
    for mode in (0, 1, 2, 0, 1, 2):
        eng.set_option("spectral_cols", mode)
        out = eng.residual(wf, ksq, src)
        for _ in range(20): eng.residual(wf, ksq, src)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): eng.residual(wf, ksq, src)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        if mode == 0: ref = out
        print(f"B={b} cols={mode}: {dt * 1e6:.1f} us per residual  bit-equal to mode 0: {torch.equal(out, ref)}")
