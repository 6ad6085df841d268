import sys, time, torch
sys.path.insert(0, "/root/repo")
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
for n, B, steps in ((512, 16, 150), (256, 32, 300)):
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
    s.set_domain_size(n, source_location=[30, n // 2])
    sos = torch.from_numpy(ring_sos_batch(n, B, seed=0)).cuda()
    eng = s.engine(); eng.reserve(B)
    k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
    st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
    rmse = torch.zeros(steps, B, device="cuda")
    eng.step(wf, res, st, k_sq, src, 300); torch.cuda.synchronize()
    out = {0: [], 1: []}
    for rep in range(12):
        for mode in ((0, 1) if rep % 2 == 0 else (1, 0)):
            eng.step(wf, res, st, k_sq, src, 30, rmse_hist=rmse[:30] if mode else None); torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.step(wf, res, st, k_sq, src, steps, rmse_hist=rmse if mode else None); torch.cuda.synchronize()
            out[mode].append((time.perf_counter() - t0) / steps * 1e3)
    med = lambda x: sorted(x)[len(x) // 2]
    print(f"{n}^2 x {B}: without rmse history {med(out[0]):.4f} ms, with {med(out[1]):.4f} ms  (+{(med(out[1]) - med(out[0])) * 1e3:.1f} us per iteration)")
