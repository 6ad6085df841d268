"""Time get_residual (hn_residual) alone: python tools/time_residual.py [N] [B] [--dense]   (--dense: the O(N^3) operator for non-power-of-two N)"""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from helmnet_amd import IterativeSolver
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 256
b = int(args[1]) if len(args) > 1 else 32
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(n, source_location=[n // 4, n // 2])
if "--dense" in sys.argv:
    s._engine = None
    from helmnet_amd.engine import Engine
    e = Engine(torch.device("cuda:0")); e.set_option("spectral_pfa", 0); s._engine = e
eng = s.engine()
wf = torch.randn(b, 2, n, n, device="cuda:0"); ksq = torch.rand(b, 1, n, n, device="cuda:0") + 0.5
src = s.source.detach().contiguous()
for _ in range(5): eng.residual(wf, ksq, src)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): eng.residual(wf, ksq, src)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
print(f"N={n} B={b} {'dense' if '--dense' in sys.argv else 'fft'}: {dt * 1e6:.1f} us per residual, {5 * 4 * n * n * b / dt / 1e9:.0f} GB/s compulsory")
