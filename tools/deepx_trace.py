"""Stage timeline of k_deepx (a -DHN_DXTRACE build: tools/build_variant.sh dxtrace hn_deepx.hip -DHN_DXTRACE):
    python tools/deepx_trace.py tools/lib_dxtrace.so [n] [batch]
100 MHz timestamps of thread 0 of every workgroup at the stage boundaries of the last launch; durations in us (mean / max over workgroups)."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.abspath(sys.argv[1])
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(n, source_location=[30, n // 2])
sos = torch.from_numpy(ring_sos_batch(n, B, seed=0)).cuda()
eng = s.engine(); eng.reserve(B)
k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
eng.step(wf, res, st, k_sq, src, 200); torch.cuda.synchronize()
lib = ctypes.CDLL(_lib._LIB_PATH)
buf = np.zeros((512, 48), dtype=np.uint64)
rc = lib.hn_debug_dx_trace(buf.ctypes.data_as(ctypes.c_void_p)); assert rc == 0, rc
nblk = 64 * ((B + 7) // 8)
t = buf[:nblk].astype(np.int64)
live = t[:, 47] > 0
t = t[live]
t0 = t[:, 0].min()
t = (t - t0) * 0.01
K2 = (t[:, 16] > 0).any()
names = {0: "start", 2: "x, state in LDS", 3: "sig conv1", 4: "sig conv2 (+ write-through)", 5: "signal out + zero inner", 6: "wait out flags", 7: "out halos in", 8: "down",
         9: "signal x'", 10: "conv_state", 11: "INNER", 12: "wait y' flags", 13: "y' halos in + zero mid", 14: "up", 15: "signal u + wait u flags", 40: "u halos in",
         41: "dec conv1", 47: "dec conv2 + store"}
inner = {16: "  start", 17: "  wait x flags", 18: "  x halos, state in", 19: "  sig conv1", 20: "  sig conv2", 21: "  signal out + zero", 22: "  wait out flags", 23: "  out halos in",
         24: "  down", 25: "  signal x'", 26: "  conv_state", 27: "  INNER", 28: "  wait y' flags", 29: "  y' halos in", 30: "  up", 31: "  signal u + wait", 43: "  u halos in",
         44: "  dec conv1", 45: "  dec conv2 + write", 46: "  signal y"}
bott = {32: "    start", 33: "    wait x flags", 34: "    x halos in", 35: "    conv1", 36: "    conv2 + write", 37: "    signal y"}
def seq(order, labels):
    prev = None
    for k in order:
        if prev is not None:
            d = t[:, k] - t[:, prev]
            print(f"{labels[k]:34s} mean {d.mean():6.2f}  max {d.max():6.2f}   (ends at mean {t[:, k].mean():6.1f})")
        prev = k
print(f"{n}^2 x {B}: {live.sum()} workgroups; kernel span {t[:, 47].max():.1f} us (first start .. last end); starts within {t[:, 0].max():.2f} us")
seq([0, 2, 3, 4, 5, 6, 7, 8, 9, 10], names)
if K2:
    seq([10, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26], {**inner, 16: "  (enter)"})
    seq([26, 32, 33, 34, 35, 36, 37], {**bott, 32: "    (enter)"})
    seq([37, 28, 29, 30, 31, 43, 44, 45, 46], {**inner, 28: "  wait y' flags"})
    seq([46, 12, 13, 14, 15, 40, 41, 47], {**names, 12: "wait y' flags"})
else:
    seq([10, 32, 33, 34, 35, 36, 37], {**bott, 32: "    (enter)"})
    seq([37, 12, 13, 14, 15, 40, 41, 47], {**names, 12: "wait y' flags"})
