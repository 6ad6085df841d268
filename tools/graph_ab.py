"""A/B: hn_step launched kernel by kernel vs replayed as HIP graphs of 1 / 2 / 8 iterations (256^2 x 32, 300 iterations)."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(256, source_location=[30, 128])
sos = torch.from_numpy(ring_sos_batch(256, 32, seed=0)).cuda()
eng = s.engine(); eng.reserve(32)
k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
rmse = torch.zeros(304, 32, device="cuda:0")
for g in (0, 1, 2, 8, 0, 1, 2, 8):
    eng.set_option("graph", g)
    eng.step(wf, res, st, k_sq, src, 32, rmse_hist=rmse[:32])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.step(wf, res, st, k_sq, src, 304, rmse_hist=rmse[:304])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"graph={g}: {304 / dt:.1f} it/s  ({dt / 304 * 1e6:.1f} us/it)  replays {eng.counter('graph_replays')} eager {eng.counter('eager_iterations')}")
