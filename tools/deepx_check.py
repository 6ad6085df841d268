"""Stage-by-stage check of hn_deepx.hip (the deep levels with eight workgroups per sample) against the layer-by-layer path.

    python tools/deepx_check.py [--n 256] [--batch 3] [--time]

Both paths leave every intermediate tensor of the fused levels in the SAME workspace buffers (the new kernel publishes every band row of its four exchange
tensors per level), so one UNet evaluation per mode and a diff per buffer names the first stage that differs: out_d (conv_signal), x_{d+1} (down),
y_{d+1} (inner result), u_d (up), y_d (decoder), and the new hidden states per level.  --time: rocprof-free timing of the loop with deep = 0 / 1 / 2.
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ws(eng, kind, level):
    lib = eng.lib
    lib.hn_debug_workspace.restype = ctypes.c_int
    lib.hn_debug_workspace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_long)]
    p, cnt = ctypes.c_void_p(), ctypes.c_long()
    assert lib.hn_debug_workspace(eng.ctx, kind, level, ctypes.byref(p), ctypes.byref(cnt)) == 0
    if not p.value:
        return None
    out = torch.empty(cnt.value, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(out.data_ptr(), p.value, cnt.value * 4, 3) == 0   # device to device
    torch.cuda.synchronize()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    from golden_inputs import teacher_inputs
    from helmnet_amd import IterativeSolver
    n, b, dev = a.n, a.batch, "cuda:0"
    ti = {k: torch.from_numpy(v).to(dev) for k, v in teacher_inputs(n, b, seed=555).items()}
    depth = 4
    res = {}
    for deep in (0, 2):
        s = IterativeSolver.from_exported_weights(); s.freeze(); s.to(dev)
        s.set_domain_size(n, source_location=[n // 8, n // 2])
        eng = s.engine()
        eng.set_option("deep", deep)
        k_sq, _ = s.get_initials(ti["sos"])
        for it in range(a.iters):    # several evaluations from the same inputs: epochs advance, results must not
            s.f.set_states(ti["states"], flatten=True)
            wf2, res2 = s.single_step(ti["wf"], k_sq, ti["res"])
            torch.cuda.synchronize()
            eng.check_async_errors()
            cur = {"wf": wf2.clone(), "res": res2.clone(), "states": s.f.get_states(flatten=True).clone()}
            for d in range(1, depth + 1):
                for kind, name in ((0, "a"), (1, "o"), (2, "y")):
                    t = ws(eng, kind, d)
                    if t is not None:
                        m = n >> d
                        cur[f"{name}{d}"] = t[: b * 8 * m * m].view(b, 8, m, m).clone()
            if it == 0:
                res[deep] = cur
            else:
                for k in cur:
                    if not torch.equal(cur[k], res[deep][k]):
                        print(f"  deep={deep}: evaluation {it} differs from evaluation 0 in {k}: {(cur[k] - res[deep][k]).abs().max().item():.3e}")
    off = 0
    print(f"n = {n}, batch = {b}: layer by layer (deep 0) vs hn_deepx (deep 2)")
    order = []
    for d in range(1, depth + 1):
        order += [f"o{d}", f"a{d + 1}", f"y{d + 1}"]
    order += [f"a{d}" for d in range(depth, 0, -1)] + [f"y{d}" for d in range(depth, 0, -1)]
    seen = set()
    for k in order:
        if k in seen or k not in res[0]:
            continue
        seen.add(k)
        x, y = res[0][k], res[2][k]
        err = (x - y).abs()
        scale = x.abs().max().item()
        bad_rows = ""
        if err.max().item() > 4e-6 * max(scale, 1e-30):
            rows = (err.amax(dim=(0, 1, 3)) > 4e-6 * scale).nonzero().flatten().tolist()
            per_b = err.amax(dim=(1, 2, 3)).tolist()
            bad_rows = f"  rows {rows[:24]}{'...' if len(rows) > 24 else ''} per sample {['%.1e' % v for v in per_b]}"
        print(f"  {k:4s} max|.| {scale:9.3e}  Linf diff {err.max().item():9.3e}  rel {err.max().item() / max(scale, 1e-30):8.1e}{bad_rows}")
    st0, st2 = res[0]["states"], res[2]["states"]
    o = 0
    for d in range(depth):
        m = n >> d
        e = (st0[:, :, o:o + m * m] - st2[:, :, o:o + m * m]).abs().max().item()
        sc = st0[:, :, o:o + m * m].abs().max().item()
        print(f"  state level {d}: Linf diff {e:9.3e} rel {e / sc:8.1e}")
        o += m * m
    for k in ("wf", "res"):
        e = (res[0][k] - res[2][k]).abs().max().item()
        print(f"  {k}: Linf diff {e:9.3e} rel {e / res[0][k].abs().max().item():8.1e}")
    if a.time:
        from helmnet_amd.phantoms import ring_sos_batch
        for nn, bb in ((256, 32), (512, 16), (256, 8), (256, 64)):
            sos = torch.from_numpy(ring_sos_batch(nn, bb, seed=3)).to(dev)
            line = []
            for deep in (0, 1, 2, 1, 2):
                s = IterativeSolver.from_exported_weights(); s.freeze(); s.to(dev)
                s.set_domain_size(nn, source_location=[nn // 8, nn // 2])
                s.engine().set_option("deep", deep)
                s.forward(sos, num_iterations=60, residuals="norms")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = s.forward(sos, num_iterations=300, residuals="norms")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                line.append(f"deep={deep}: {300 / dt:7.1f} it/s (rmse {o['residual_norms'][-1].max().item():.2e})")
                s.engine().check_async_errors()
            print(f"  {nn}^2 x {bb}: " + "   ".join(line))


if __name__ == "__main__":
    main()
