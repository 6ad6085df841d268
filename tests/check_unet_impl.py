"""Parity of one UNet precision mode (hn_set_unet_precision; per context, so every mode runs in the SAME process):
single_step against the CPU oracle on seeded inputs, the network output against the oracle and against a float64
evaluation, and the config-1 / README free runs against the reference's committed golden traces.

    python tests/check_unet_impl.py bf16x3          # prints one JSON line
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):   # test infrastructure: may use the oracle
    if p not in sys.path:
        sys.path.insert(0, p)


def check(impl: str, dev="cuda:0") -> dict:
    from golden_inputs import teacher_inputs
    from helmnet_amd import HybridNet, IterativeSolver
    from helmnet_amd.phantoms import readme_sos
    from oracle import helmnet_oracle as O

    dev = torch.device(dev)
    s = IterativeSolver.from_exported_weights()
    s.freeze()
    s.to(dev)
    s.set_unet_precision(impl)
    with np.load(os.path.join(ROOT, "tests", "golden", "jcp_weights.npz")) as z:
        weights = {k: torch.from_numpy(z[k]) for k in z.files}
    out = {"impl": impl}
    for n, b in ((256, 2), (128, 2)):
        ti = {k: torch.from_numpy(v) for k, v in teacher_inputs(n, b, seed=77 + n).items()}
        loc = [n // 4, n // 2]
        s.set_domain_size(n, source_location=loc)
        t = O.SpectralTables(n, 8, 2, 1.0)
        src = O.point_source_map(n, loc, 10.0)
        k_sq_o, _ = O.get_initials(ti["sos"], 1.0)
        st = O.unflatten_states(ti["states"], n, 4)
        want_wf, want_res, want_st = O.single_step(ti["wf"], k_sq_o, ti["res"], st, weights, src, t)
        g = {k: v.to(dev) for k, v in ti.items()}
        k_sq, _ = s.get_initials(g["sos"])
        s.f.set_states(g["states"], flatten=True)
        wf2, res2 = s.single_step(g["wf"], k_sq, g["res"])
        assert s.engine().unet_precision == impl
        out[f"single_step_{n}"] = {
            "wf": ((wf2.cpu() - want_wf).abs().max() / want_wf.abs().max()).item(),
            "res": ((res2.cpu() - want_res).abs().max() / want_res.abs().max()).item(),
        }
    # the network output itself (single_step's wavefield error is dominated by the rounding of wf + d/1e3)
    for n in (256, 128):
        net = HybridNet("prelu", 4, n, 8, 6, 2, 4)
        net.load_state_dict(weights)
        net.to(dev)
        net.set_unet_precision(impl)
        x = torch.randn(2, 6, n, n, generator=torch.Generator().manual_seed(11 + n))
        net.clear_states(x.to(dev))
        d = net(x.to(dev)).cpu()
        want, _ = O.unet_forward(x, [torch.zeros(2, 2, m, m) for m in O.state_dims(n, 4)], weights)
        out[f"unet_output_{n}"] = ((d - want).abs().max() / want.abs().max()).item()
        # against a float64 evaluation of the same network: how far each implementation (and the fp32 CPU oracle
        # itself) is from the exact answer
        w64 = {k: v.double() for k, v in weights.items()}
        truth, _ = O.unet_forward(x.double(), [torch.zeros(2, 2, m, m, dtype=torch.float64) for m in O.state_dims(n, 4)], w64)
        out[f"unet_output_{n}_vs_fp64"] = ((d.double() - truth).abs().max() / truth.abs().max()).item()
        out[f"unet_output_{n}_vs_fp64_rms"] = ((d.double() - truth).pow(2).mean().sqrt() / truth.pow(2).mean().sqrt()).item()
        out[f"oracle_fp32_{n}_vs_fp64"] = ((want.double() - truth).abs().max() / truth.abs().max()).item()
    with np.load(os.path.join(ROOT, "tests", "golden", "free_run.npz")) as z:
        s.set_domain_size(256, source_location=[30, 128])
        o = s.forward(torch.ones(1, 1, 256, 256, device=dev), num_iterations=100)
        out["cfg1_wf_linf_vs_reference"] = float(np.abs(o["wavefields"][0].cpu().numpy() - z["cfg1_wf_it100"]).max())
        out["cfg1_rmse_rel"] = float(np.abs(o["residual_norms"].cpu().numpy() / z["cfg1_rmse"] - 1).max())
        # README problem, 300 iterations: converged residual and wavefield against the reference's own run
        o = s.forward(torch.from_numpy(readme_sos()).to(dev), num_iterations=300, residuals="norms")
        rm = o["residual_norms"].cpu().numpy()
        out["readme300_final_rmse"] = float(rm[-1].max())
        out["readme300_final_rmse_reference"] = float(z["readme_rmse"][-1].max())
        out["readme300_rmse_rel"] = float(np.abs(rm / z["readme_rmse"] - 1).max())
        out["readme300_wf_linf_vs_reference"] = float(np.abs(o["wavefields"][0].cpu().numpy() - z["readme_wf_it300"]).max())
    return out


if __name__ == "__main__":
    print(json.dumps(check(sys.argv[1] if len(sys.argv) > 1 else "fp32")))
