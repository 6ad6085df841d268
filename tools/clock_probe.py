"""Effective shader clock while the solver loop runs (DVFS): a one-wave probe kernel on a second stream samples
s_memtime against the 100 MHz s_memrealtime in 20 us windows.   python tools/clock_probe.py [precision]"""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch
SO = os.path.join(ROOT, "tools", "probe", "libclock_probe.so")
if not os.path.exists(SO):   # git-ignored: build it here (hipcc cross-compiles, so this works before the file travels to the GPU box)
    import subprocess
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO,
                    os.path.join(ROOT, "tools", "probe", "clock_probe.hip")], check=True)
lib = ctypes.CDLL(SO)
lib.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong]
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
if len(sys.argv) > 1: s.set_unet_precision(sys.argv[1])
s.set_domain_size(256, source_location=[30, 128])
sos = torch.from_numpy(ring_sos_batch(256, 32, seed=0)).cuda()
eng = s.engine(); eng.reserve(32)
k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
N = 2000
out = torch.zeros(2 * N, dtype=torch.int64, device="cuda:0")
side = torch.cuda.Stream()
def probe(tag, busy):
    out.zero_(); torch.cuda.synchronize()
    if busy: eng.step(wf, res, st, k_sq, src, 100)           # ~57 ms of solver iterations ...
    lib.probe_launch(ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(out.data_ptr()), N, 2000)   # ... 2000 windows of 20 us = 40 ms beside them
    torch.cuda.synchronize()
    o = out.cpu().view(N, 2).double()
    mhz = o[:, 0] / o[:, 1] * 100.0
    print(f"{tag}: shader clock median {mhz.median():.0f} MHz, 5 % .. 95 %: {mhz.quantile(0.05):.0f} .. {mhz.quantile(0.95):.0f} MHz")
eng.step(wf, res, st, k_sq, src, 200); torch.cuda.synchronize()
probe("idle GPU (probe alone)", False)
for _ in range(3): probe("solver loop running", True)
