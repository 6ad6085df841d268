"""What do the hidden-state kernels on the side stream cost the main chain?  In-process: the loop with conv_state_e launched (1) or skipped (0) -- results of the
skipping runs are NOT the solver's.  Needs tools/lib_repeat.so.   python tools/cs_skip_probe.py [--size 512 --batch 16]"""
import argparse, ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.join(ROOT, "tools", "lib_repeat.so")
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--steps", type=int, default=150)
a = ap.parse_args()
n, B = a.size, a.batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(n, source_location=[30, n // 2])
sos = torch.from_numpy(ring_sos_batch(n, B, seed=0)).cuda()
eng = s.engine(); eng.reserve(B)
lib = eng.lib
lib.hn_debug_set_repeat.argtypes = [ctypes.c_int, ctypes.c_int]; lib.hn_debug_set_repeat.restype = None
k_sq, wf0 = s.get_initials(sos); s.f.clear_states(wf0); res0 = s.get_residual(wf0, k_sq)
st0 = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
eng.step(wf0, res0, st0, k_sq, src, 40); torch.cuda.synchronize()
KID_STATE0 = 2   # conv_state_e = 2 + 3 e (hn_internal.h)
assert eng.kernel_name(KID_STATE0) == "conv_state0"
masks = [0, 1, 2, 4, 8, 15]
out = {m: [] for m in masks}
for rep in range(7):
    for m in (masks if rep % 2 == 0 else masks[::-1]):
        for e in range(4): lib.hn_debug_set_repeat(KID_STATE0 + 3 * e, 0 if (m >> e) & 1 else 1)
        wf, res, st = wf0.clone(), res0.clone(), st0.clone()
        eng.step(wf, res, st, k_sq, src, 20); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.step(wf, res, st, k_sq, src, a.steps); torch.cuda.synchronize()
        out[m].append((time.perf_counter() - t0) / a.steps * 1e6)
med = lambda x: sorted(x)[len(x) // 2]
base = med(out[0])
print(f"{n}^2 x {B}: all hidden-state kernels launched: {base:.1f} us per iteration")
for m in masks[1:]:
    print(f"  skipping levels {[e for e in range(4) if (m >> e) & 1]}: {med(out[m]):.1f} us  ({med(out[m]) - base:+.1f})")
