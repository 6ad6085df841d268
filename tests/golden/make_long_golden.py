#!/usr/bin/env python3
"""Long free-run fixtures: the REFERENCE run for the full
iteration counts of BASELINE.json configs[1], [3] and [4], in float64 and in its native float32.

Runs only in the build container (imports the unmodified reference through the stand-ins of
make_golden.py).  float64 reference = the same module after ``.double()`` under
``torch.set_default_dtype(torch.float64)`` (``get_initials`` / ``clear_state`` create default-dtype
zeros, hybridnet.py:535-537, architectures.py:236-238): the SAME fp32-rounded constants and weights,
evaluated in exact-enough arithmetic.  It anchors the long-run bound: an fp32 implementation may drift
from it by about as much as the reference's own fp32 run does.

    python tests/golden/make_long_golden.py [cfg2] [cfg4] [cfg5]      # -> tests/golden/long_run.npz

Stored (data only): strided wavefield probes of the float64 trajectory at checkpoints, the per-iteration
per-sample residual RMSE of both runs, and the reference-fp32 deviation from float64 (full-field and
probe-only L-infinity) at every checkpoint.  Inputs are re-created from seeds (tests/golden_inputs.py).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, npf  # noqa: E402

from golden_inputs import long_inputs  # noqa: E402  (tests/ is on sys.path via make_golden)

OUT = os.path.join(HERE, "long_run.npz")


def run(solver, sos, n, iters, keep_at, stride, loc=None, src_map=None, dtype=torch.float32):
    torch.set_default_dtype(dtype)
    try:
        if loc is not None:
            solver.set_domain_size(n, source_location=loc)
        else:
            solver.set_domain_size(n, source_map=src_map.to(dtype))
        solver.to(dtype)
        solver.sigmas = solver.sigmas.to(dtype)
        sos = sos.to(dtype)
        k_sq, wf = solver.get_initials(sos)
        solver.f.clear_states(wf)
        res = solver.get_residual(wf, k_sq)
        assert wf.dtype == dtype and res.dtype == dtype, (wf.dtype, res.dtype)
        trace, keep = [], {}
        t0 = time.time()
        for it in range(iters):
            wf, res = solver.single_step(wf, k_sq, res)
            assert wf.dtype == dtype
            trace.append(solver.test_loss_function(res).double().numpy())
            if it + 1 in keep_at:
                keep[it + 1] = wf.double().clone()
            if (it + 1) % 250 == 0:
                print(f"   it {it + 1}/{iters}  rmse max {trace[-1].max():.3e}  ({time.time() - t0:.0f} s)", flush=True)
        return np.stack(trace), keep
    finally:
        torch.set_default_dtype(torch.float32)
        solver.float()


def case(solver, out, tag, sos, n, iters, keep_at, stride, both=True, **kw):
    print(f"== {tag}: {tuple(sos.shape)} {iters} it", flush=True)
    r32, k32 = run(solver, sos, n, iters, keep_at, stride, dtype=torch.float32, **kw)
    out[f"{tag}_rmse_f32"] = r32.astype(np.float32)
    if both:
        r64, k64 = run(solver, sos, n, iters, keep_at, stride, dtype=torch.float64, **kw)
        out[f"{tag}_rmse_f64"] = r64.astype(np.float32)
    for it in keep_at:
        ref = k64[it] if both else k32[it]
        out[f"{tag}_wf_it{it}"] = ref[:, :, ::stride, ::stride].numpy().astype(np.float32)
        out[f"{tag}_wf_absmax_it{it}"] = ref.abs().amax((1, 2, 3)).numpy().astype(np.float32)
        if both:
            d = (k32[it] - k64[it]).abs()
            out[f"{tag}_f32dev_it{it}"] = d.amax((1, 2, 3)).numpy().astype(np.float32)
            out[f"{tag}_f32dev_probe_it{it}"] = d[:, :, ::stride, ::stride].amax((1, 2, 3)).numpy().astype(np.float32)
            print(f"   it {it}: reference fp32 vs fp64 Linf per sample {out[f'{tag}_f32dev_it{it}']}", flush=True)
    out[f"{tag}_stride"] = np.int32(stride)


def main():
    which = set(sys.argv[1:]) or {"cfg2", "cfg4", "cfg5"}
    torch.manual_seed(0)
    torch.set_num_threads(8)
    solver = load_reference()
    out = {}
    if os.path.exists(OUT):
        with np.load(OUT) as z:
            out = {k: z[k] for k in z.files}
    with torch.no_grad():
        if "cfg2" in which:   # configs[1]: 256^2, 1000 iterations (README map + 4 ring maps)
            li = long_inputs("cfg2")
            case(solver, out, "cfg2", torch.from_numpy(li["sos"]), 256, 1000, (100, 300, 1000), 3, loc=li["loc"])
        if "cfg4" in which:   # configs[3]: 512^2 via set_domain_size, 2000 iterations
            li = long_inputs("cfg4")
            case(solver, out, "cfg4", torch.from_numpy(li["sos"]), 512, 2000, (500, 1000, 2000), 4, loc=li["loc"])
        if "cfg5" in which:   # configs[4]: synthetic skull + arc source map, up to 3000 iterations (fp32 only)
            li = long_inputs("cfg5")
            case(solver, out, "cfg5", torch.from_numpy(li["sos"]), 512, 3000, (1000, 2000, 3000), 4, both=False,
                 src_map=torch.from_numpy(li["src_map"]))
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
