"""In-process A/B of a library option on the solver loop: ONE solver, the option flipped between hn_step calls, many alternations.

    python tools/ab_inproc.py side_sync 0,1,2 [--size 256 --batch 32 --steps 300 --reps 15] [--clock]

Box-to-box (and run-to-run) spread of the headline is +-2 %; effects of 1 % only show when the variants alternate in one process under the same clocks and
temperature.  Prints median / quartiles of ms per iteration per value and the ratio to the first value; --clock adds the median shader clock and board power
(hwmon) sampled while each variant runs."""
import argparse, glob, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helmnet_amd import _lib, IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch

ap = argparse.ArgumentParser()
ap.add_argument("option"); ap.add_argument("values")
ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=300); ap.add_argument("--reps", type=int, default=15)
ap.add_argument("--clock", action="store_true")
ap.add_argument("--lib", default=None, help="an alternative build of the library (tools/build_variant.sh)")
a = ap.parse_args()
vals = [int(v) for v in a.values.split(",")]
if a.lib:
    _lib._LIB_PATH = os.path.abspath(a.lib)

def find(pattern):
    for p in sorted(glob.glob(pattern)):
        try:
            int(open(p).read().split()[0]); return p
        except Exception:
            pass
    return None
P = F = None
if a.clock:
    pr = torch.cuda.get_device_properties(0)
    addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0" if hasattr(pr, "pci_bus_id") else None
    hw = f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/" if addr and glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/") else "/sys/class/drm/card*/device/hwmon/hwmon*/"
    P = find(hw + "power1_average") or find(hw + "power1_input"); F = find(hw + "freq1_input")

n, B = a.size, a.batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(n, source_location=[30, n // 2])
sos = torch.from_numpy(ring_sos_batch(n, B, seed=0)).cuda()
eng = s.engine(); eng.reserve(B)
k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
eng.step(wf, res, st, k_sq, src, 600); torch.cuda.synchronize()   # clocks and caches settle
ms = {v: [] for v in vals}; clk = {v: [] for v in vals}; pw = {v: [] for v in vals}
def sample(stop, acc):
    while not stop.is_set():
        try: acc.append((int(open(P).read()) / 1e6 if P else 0.0, int(open(F).read()) / 1e6 if F else 0.0))
        except Exception: pass
        time.sleep(0.004)
for rep in range(a.reps):
    order = vals if rep % 2 == 0 else vals[::-1]
    for v in order:
        eng.set_option(a.option, v)
        eng.step(wf, res, st, k_sq, src, 40); torch.cuda.synchronize()
        stop, acc = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, acc)) if a.clock else None
        if th: th.start()
        t0 = time.perf_counter()
        eng.step(wf, res, st, k_sq, src, a.steps); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if th: stop.set(); th.join()
        ms[v].append(dt / a.steps * 1e3)
        if acc:
            acc.sort(key=lambda t: t[1]); clk[v].append(acc[len(acc) // 2][1]); acc.sort(); pw[v].append(acc[len(acc) // 2][0])
q = lambda x, f: sorted(x)[min(len(x) - 1, int(f * len(x)))]
base = q(ms[vals[0]], 0.5)
print(f"{a.option} on {n}^2 x {B}, {a.reps} alternations of {a.steps} iterations (finite wavefield: {bool(torch.isfinite(wf).all())})")
for v in vals:
    m = q(ms[v], 0.5)
    extra = f"   sclk {q(clk[v], 0.5):.0f} MHz, {q(pw[v], 0.5):.0f} W" if clk[v] else ""
    print(f"  {a.option}={v}: median {m:.4f} ms  (q1 {q(ms[v], 0.25):.4f}, q3 {q(ms[v], 0.75):.4f}, min {min(ms[v]):.4f})  {1e3 / m:7.1f} it/s   x{base / m:.4f} vs {a.option}={vals[0]}{extra}")
