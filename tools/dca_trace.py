"""Phase timeline of the decoder instance of k_dc_asm (a -DHN_ATRACE build: tools/build_variant.sh trace hn_dca.hip -DHN_ATRACE):
python tools/dca_trace.py tools/lib_trace.so"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.abspath(sys.argv[1])
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(256, source_location=[30, 128])
sos = torch.from_numpy(ring_sos_batch(256, 32, seed=0)).cuda()
eng = s.engine(); eng.reserve(32); eng.set_option("dc_valu", int(sys.argv[2]) if len(sys.argv) > 2 else 4)
k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
eng.step(wf, res, st, k_sq, src, 300); torch.cuda.synchronize()
lib = ctypes.CDLL(_lib._LIB_PATH)
buf = np.zeros((8192, 8), dtype=np.uint64)
rc = lib.hn_debug_dca_trace(buf.ctypes.data_as(ctypes.c_void_p)); assert rc == 0, rc
t = buf[:2048, :7].astype(np.int64); hw = buf[:2048, 7]
t0 = t[:, 0].min(); t = (t - t0) * 0.01   # us
print("kernel span (first block start .. last block end): %.1f us" % (t[:, 6].max()))
names = ["plan+first issue", "conv1 loop", "barrier A", "exchange+finish | put half 0", "conv2 setup+barrier | conv2 half 0", "conv2 loop | put half 1 + conv2 half 1", ]
for i, nme in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print(f"  {nme:22s} mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p95 {np.percentile(d, 95):6.2f}")
life = t[:, 6] - t[:, 0]
print("  block lifetime (to conv2 end) mean %.2f median %.2f min %.2f max %.2f" % (life.mean(), np.median(life), life.min(), life.max()))
st_ = np.sort(t[:, 0]); print("  block start times: first 1024 by %.2f us; 1025th at %.2f; last at %.2f" % (st_[1023], st_[1024], st_[-1]))
cu = ((hw >> 8) & 0xf).astype(int) + 16 * ((hw >> 13) & 0x7).astype(int) + 128 * ((hw >> 16) & 0xf).astype(int)   # cu_id, se_id, (xcc?)
print("  distinct (cu, se, ...) ids:", len(set(cu.tolist())), " blocks per id: min %d max %d" % (np.bincount(cu).min(), np.bincount(cu).max()))
# per early-vs-late start
early = t[:, 0] < st_[1023] + 0.01
for tag, m in (("first-round blocks", early), ("second-round blocks", ~early)):
    if m.sum():
        print(f"  {tag}: n={m.sum()} conv1 {np.mean(t[m,2]-t[m,1]):.2f} us, exchange {np.mean(t[m,4]-t[m,3]):.2f}, conv2 {np.mean(t[m,6]-t[m,5]):.2f}, lifetime {np.mean(t[m,6]-t[m,0]):.2f}, end at mean {np.mean(t[m,6]):.1f} max {np.max(t[m,6]):.1f}")
