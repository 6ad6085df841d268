"""In-process A/B of HN_OPT_SIDE_SYNC on the training step (96^2 x 32 x 10): gradient bit-identity and time."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from helmnet_amd import IterativeSolver
from helmnet_amd.engine import pack_weights
from helmnet_amd.phantoms import ring_sos_batch
n, B, unroll = 96, 32, 10
s = IterativeSolver.from_exported_weights(); s.to("cuda:0")
s.set_domain_size(n, source_location=[n - 14, n // 2])
sos = torch.from_numpy(ring_sos_batch(n, B, seed=5)).cuda()
eng = s.engine(); eng.reserve(B)
k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
eng.step(wf, res, st, k_sq, src, 5)
src_b = src.repeat(B, 1, 1, 1).contiguous()
w = torch.from_numpy(pack_weights(dict(s.f.state_dict()))).cuda()
grads = {}
for v in (1, 0):
    eng.set_option("side_sync", v)
    g = torch.zeros_like(w)
    o = eng.train_grad(w, wf, res, st, k_sq, src_b, unroll, 1e4, grad=g); torch.cuda.synchronize()
    grads[v] = (g.clone(), float(o["loss"][0]))
print("gradients bit-identical:", torch.equal(grads[1][0], grads[0][0]), " loss equal:", grads[1][1] == grads[0][1], grads[1][1])
ms = {0: [], 1: []}
for rep in range(9):
    for v in ((0, 1) if rep % 2 == 0 else (1, 0)):
        eng.set_option("side_sync", v)
        g = torch.zeros_like(w)
        for _ in range(3): eng.train_grad(w, wf, res, st, k_sq, src_b, unroll, 1e4, grad=g)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): eng.train_grad(w, wf, res, st, k_sq, src_b, unroll, 1e4, grad=g)
        torch.cuda.synchronize(); ms[v].append((time.perf_counter() - t0) / 8 * 1e3)
med = lambda x: sorted(x)[len(x) // 2]
print(f"hn_train_grad: side_sync 0: {med(ms[0]):.3f} ms, 1: {med(ms[1]):.3f} ms  (x{med(ms[0]) / med(ms[1]):.4f})")
