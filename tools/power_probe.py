"""Board power and shader clock (sysfs hwmon, ~5 ms sampling) while the solver loop runs:  python tools/power_probe.py [dc_valu values ...]"""
import glob, os, sys, threading, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helmnet_amd import IterativeSolver
from helmnet_amd.phantoms import ring_sos_batch

def find(pattern):
    for p in sorted(glob.glob(pattern)):
        try:
            int(open(p).read().split()[0]); return p
        except Exception:
            pass
    return None
# the hwmon directory of the GPU torch runs on (a box can expose several cards): matched by PCI address
pr = torch.cuda.get_device_properties(0)
addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0" if hasattr(pr, "pci_bus_id") else None
hw = f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/" if addr and glob.glob(f"/sys/bus/pci/devices/{addr}/hwmon/hwmon*/") else "/sys/class/drm/card*/device/hwmon/hwmon*/"
P = find(hw + "power1_average") or find(hw + "power1_input")
F = find(hw + "freq1_input")
CAP = find(hw + "power1_cap")
print("power file", P, "freq file", F, "cap", open(CAP).read().strip() if CAP else None, flush=True)
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
s.set_domain_size(256, source_location=[30, 128])
sos = torch.from_numpy(ring_sos_batch(256, 32, seed=0)).cuda()
eng = s.engine(); eng.reserve(32)
k_sq, wf = s.get_initials(sos); s.f.clear_states(wf); res = s.get_residual(wf, k_sq)
st = s.f.get_states(flatten=True).contiguous(); k_sq = k_sq.contiguous(); src = s.source.detach().contiguous()
def sample(stop, acc):
    while not stop.is_set():
        try:
            acc.append((time.perf_counter(), int(open(P).read()) / 1e6 if P else 0.0, int(open(F).read()) / 1e6 if F else 0.0))
        except Exception:
            pass
        time.sleep(0.004)
for v in [int(a) for a in sys.argv[1:]] or [0, 1, 4]:
    eng.set_option("dc_valu", v)
    eng.step(wf, res, st, k_sq, src, 200); torch.cuda.synchronize()
    stop, acc = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, acc)); th.start()
    t0 = time.perf_counter()
    eng.step(wf, res, st, k_sq, src, 3000); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    acc = [a for a in acc if a[0] - t0 > 0.3]
    pw = sorted(a[1] for a in acc); fq = sorted(a[2] for a in acc)
    med = lambda x: x[len(x) // 2] if x else 0
    print(f"dc_valu={v}: {3000 / dt:.1f} it/s; power median {med(pw):.0f} W (min {pw[0] if pw else 0:.0f}, max {pw[-1] if pw else 0:.0f}); sclk median {med(fq):.0f} MHz (min {fq[0] if fq else 0:.0f}); {len(acc)} samples", flush=True)
