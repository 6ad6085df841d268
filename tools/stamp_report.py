"""Diagnostic: per-phase cycle shares of the persistent DoubleConv (needs tools/libhelmnet_stamp.so,
a -DHN_STAMP build).  Prints median cycles per phase for the decoder kernel at B=32, N=256."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhelmnet_stamp.so")
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0"); s.set_domain_size(256, source_location=[30, 128])
sos = torch.from_numpy(ring_sos_batch(256, 32, seed=0)).cuda()
lib0 = _lib.load()
sel = int(sys.argv[1]) if len(sys.argv) > 1 else 881
wsel = int(sys.argv[2]) if len(sys.argv) > 2 else 256
print("stamp selector", sel, "width", wsel, "rc", lib0.hn_debug_set_stamp_sel(sel), lib0.hn_debug_set_stamp_w(wsel))
out = s.forward(sos, num_iterations=3, residuals="norms")
torch.cuda.synchronize()
lib = _lib.load()
n = 256 * 4 * 64
buf = (ctypes.c_ulonglong * n)()
lib.hn_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.hn_debug_read_stamps(buf, n))
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 64).astype(np.int64)
names = ["commit", "barrier1", "issue_next", "conv1", "mid", "barrier2", "conv2", "epilogue"]
ntile = int(sys.argv[3]) if len(sys.argv) > 3 else 7
for w in range(4):
    d = []
    for t in range(ntile):
        st = a[:, w, t * 9:(t + 1) * 9]
        d.append(np.diff(st, axis=1))
    d = np.stack(d)  # tiles, blocks, phases
    med = np.median(d.reshape(-1, 8), axis=0)
    print("wave", w, {k: int(v) for k, v in zip(names, med)}, "sum", int(med.sum()))
tile_gap = np.median(a[:, 0, 9] - a[:, 0, 0]); print("tile period (cycles @100MHz-scaled?)", int(tile_gap))
