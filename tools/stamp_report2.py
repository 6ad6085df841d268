"""Diagnostic: per-phase cycle shares of the strip-mapped DoubleConv (tools/libhelmnet_stamp.so, -DHN_STAMP)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhelmnet_stamp.so")
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
lib = _lib.load()
sel = int(sys.argv[1]) if len(sys.argv) > 1 else 881
lib.hn_debug_set_stamp_sel(sel)
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0"); s.set_domain_size(256, source_location=[30, 128])
sos = torch.from_numpy(ring_sos_batch(256, 32, seed=0)).cuda()
s.forward(sos, num_iterations=3, residuals="norms"); torch.cuda.synchronize()
n = 2048 * 4 * 16
buf = (ctypes.c_ulonglong * n)()
lib.hn_debug_read_stamps2.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.hn_debug_read_stamps2(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 4, 16).astype(np.int64)
names = ["prologue", "commit+loadwait", "chunk_barrier", "fetch_issue", "conv1_mfma", "mid_write", "mid_barrier", "conv2", "epilogue"]
med = np.median(a[:, :, :9].reshape(-1, 9), axis=0)
print("selector", sel, {k: int(v) for k, v in zip(names, med)}, "sum", int(med.sum()))
t0 = a[:, 0, 15]; order = np.argsort(t0); life = a[:, :, :9].sum(axis=2).mean(axis=1)
print("block lifetime median", int(np.median(life)), "kernel span", int(t0.max() - t0.min() + np.median(life)))
# block timeline (10 ns ticks of s_memrealtime): when do blocks start / end relative to the first start
end = t0 + a[:, :, :9].sum(axis=2).max(axis=1)
z = t0.min()
s_rel, e_rel = np.sort(t0 - z), np.sort(end - z)
pct = [0, 10, 25, 50, 75, 90, 100]
print("block start (us) pct", pct, [round(float(np.percentile(s_rel, p)) / 100, 1) for p in pct])
print("block end   (us) pct", pct, [round(float(np.percentile(e_rel, p)) / 100, 1) for p in pct])
nb = len(t0)
first = np.sort(t0 - z)[: nb // 2]
print("first-round (first half of blocks) start span us", round(float(first.max()) / 100, 1))
# concurrency over time
ts = np.arange(0, int(e_rel.max()) + 1, 100)
conc = [(int(((t0 - z) <= t).sum() - ((end - z) <= t).sum())) for t in ts]
print("resident blocks every 1 us:", conc)
