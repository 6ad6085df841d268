"""Board energy and time per launch of every kernel of the iteration, by difference (needs tools/lib_repeat.so: bash tools/build_repeat_variant.sh).

    python tools/energy_probe.py [--size 256 --batch 32 --repeat 4 --seconds 1.2]

The solver loop sits on a board-power plateau (~1100 W on the boxes of this pool: shader clock 2.29-2.32 GHz instead of 2.4) -- what a kernel costs the
loop is then its ENERGY, not only its time.  For each kernel id k the loop runs with k launched `repeat` times per iteration (same arguments; the results are not
the solver's); against the plain loop:   time per launch = (T_R - T_1) / (R - 1),   energy per launch = (P_R T_R - P_1 T_1) / (R - 1),
with T the time per iteration and P the median board power (hwmon, ~4 ms sampling) while the loop runs.  Idle board power is subtracted nowhere: a launch's
energy includes the chip's static power for its duration."""
import argparse, ctypes, glob, os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helmnet_amd import _lib
_lib._LIB_PATH = os.path.join(ROOT, "tools", "lib_repeat.so")
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--repeat", type=int, default=4); ap.add_argument("--seconds", type=float, default=1.2)
a = ap.parse_args()

def find(pattern):
    for p in sorted(glob.glob(pattern)):
        try:
            int(open(p).read().split()[0]); return p
        except Exception:
            pass
    return None
pr = torch.cuda.get_device_properties(0)
addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0" if hasattr(pr, "pci_bus_id") else None
hw = f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/" if addr and glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/") else "/sys/class/drm/card*/device/hwmon/hwmon*/"
P = find(hw + "power1_average") or find(hw + "power1_input"); F = find(hw + "freq1_input")

n, B = a.size, a.batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(n, source_location=[30, n // 2])
sos = torch.from_numpy(ring_sos_batch(n, B, seed=0)).cuda()
eng = s.engine(); eng.reserve(B)
lib = eng.lib
lib.hn_debug_set_repeat.argtypes = [ctypes.c_int, ctypes.c_int]; lib.hn_debug_set_repeat.restype = None
k_sq, wf0 = s.get_initials(sos); s.f.clear_states(wf0); res0 = s.get_residual(wf0, k_sq)
st0 = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
wf1, res1, st1 = wf0.clone(), res0.clone(), st0.clone()

def sample(stop, acc):
    while not stop.is_set():
        try: acc.append((int(open(P).read()) / 1e6 if P else 0.0, int(open(F).read()) / 1e6 if F else 0.0))
        except Exception: pass
        time.sleep(0.004)
def run(its, chunk=60):
    # (repeated launches of the final layer add their update several times and the loop diverges within a few hundred iterations -- NaN data draws far less
    # power: every chunk of iterations starts from the same early state, and the run says whether its wavefield stayed finite)
    wf, res, st = wf1.clone(), res1.clone(), st1.clone()
    eng.step(wf, res, st, k_sq, src, chunk); torch.cuda.synchronize()
    stop, acc = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, acc)); th.start()
    busy, done, finite = 0.0, 0, True
    while done < its:
        wf.copy_(wf1); res.copy_(res1); st.copy_(st1); torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.step(wf, res, st, k_sq, src, chunk); torch.cuda.synchronize()
        busy += time.perf_counter() - t0
        done += chunk
    finite = bool(torch.isfinite(wf).all())
    stop.set(); th.join()
    acc = acc[len(acc) // 4:]   # (the first quarter: the power reading lags)
    pw = sorted(x[0] for x in acc); fq = sorted(x[1] for x in acc)
    return busy / done, pw[len(pw) // 2], fq[len(fq) // 2], finite
its = int(a.seconds / 0.0005)
eng.step(wf1, res1, st1, k_sq, src, 40); torch.cuda.synchronize()   # (the state every chunk starts from: 40 iterations in)
run(its)
T1, P1, F1, ok1 = run(its)
print(f"{n}^2 x {B}: plain loop {T1 * 1e6:.1f} us / iteration, {P1:.0f} W, sclk {F1:.0f} MHz  ->  {T1 * P1 * 1e3:.1f} mJ / iteration", flush=True)
ids = [k for k in range(eng.KERNEL_IDS)]
rows = []
probe = {}
eng.profile_enable(None); eng.step(wf0.clone(), res0.clone(), st0.clone(), k_sq, src, 2); torch.cuda.synchronize(); launched = eng.profile_collect(); eng.profile_enable([])
names = [k for k in launched if k not in ("spectral_cols", "spectral_rows")]
R = a.repeat
for k in ids:
    name = eng.kernel_name(k)
    if name not in names: continue
    lib.hn_debug_set_repeat(k, R)
    T, Pw, Fq, ok = run(int(its * 0.8))
    lib.hn_debug_set_repeat(k, 1)
    t_k = (T - T1) / (R - 1); e_k = (Pw * T - P1 * T1) / (R - 1)
    rows.append((name, t_k, e_k, Pw, Fq))
    print(f"  {name:18s} x{R}: {T * 1e6:7.1f} us / iteration, {Pw:5.0f} W, sclk {Fq:4.0f} MHz   ->  per launch {t_k * 1e6:6.1f} us, {e_k * 1e3:6.2f} mJ  ({e_k / max(t_k, 1e-9):5.0f} W while it runs){'' if ok else '   [wavefield not finite: INVALID]'}", flush=True)
T1b, P1b, F1b, _ = run(its)
print(f"plain loop again: {T1b * 1e6:.1f} us, {P1b:.0f} W, {F1b:.0f} MHz")
main = [r for r in rows if not r[0].startswith("conv_state")]
print(f"sum over the main chain: {sum(r[1] for r in main) * 1e6:.1f} us, {sum(r[2] for r in main) * 1e3:.1f} mJ;  side stream (conv_state): {sum(r[2] for r in rows if r[0].startswith('conv_state')) * 1e3:.1f} mJ;  "
      f"loop: {T1 * 1e6:.1f} us, {T1 * P1 * 1e3:.1f} mJ")
