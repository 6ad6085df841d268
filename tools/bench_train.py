"""Time one training step (hn_train_grad + hn_adam_step) at the reference's training shape: 96^2, batch 32, 10 unrolled iterations
(hybridnet.py:385-413; hparams batch_size 32, unrolling_steps 10).  Usage: python tools/bench_train.py [--n 96] [--batch 32] [--unroll 10]
[--steps 10] [--cpu] (--cpu: also time the oracle's autograd on the host, the reference's own arithmetic)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--unroll", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--side-stream", action="store_true", help="run on a torch side stream instead of the default (null) stream")
    ap.add_argument("--lanes", type=int, default=None, help="HN_OPT_TRAIN_LANES (1 or 2; default: the library's)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="hn_set_option (e.g. train_fused=0)")
    a = ap.parse_args()
    from helmnet_amd import IterativeSolver
    from helmnet_amd.engine import pack_weights
    from helmnet_amd.phantoms import ring_sos_batch
    dev = "cuda:0"
    s = IterativeSolver.from_exported_weights()
    s.to(dev)
    loc = [a.n - 14, a.n // 2]
    s.set_domain_size(a.n, source_location=loc)
    eng = s.engine()
    if a.lanes is not None:
        eng.set_option("train_lanes", a.lanes)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    sos = torch.from_numpy(ring_sos_batch(a.n, a.batch, seed=5)).to(dev)
    out = s.forward(sos, num_iterations=5, return_wavefields=True, return_states=True)
    wf, res, st = out["wavefields"][-1].contiguous(), out["residuals"][-1].contiguous(), out["states"][-1].contiguous()
    k_sq = ((1.0 / sos) ** 2).contiguous()
    src = s.source.detach().repeat(a.batch, 1, 1, 1).contiguous()
    w = torch.from_numpy(pack_weights(dict(s.f.state_dict()))).to(dev)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    g = torch.zeros_like(w)

    def step(i):
        o = eng.train_grad(w, wf, res, st, k_sq, src, a.unroll, 1e4, grad=g)
        eng.adam_step(w, g, m, v, i + 1, 1e-5, (0.9, 0.95), 1e-8, 1e-6, 1.0)
        return o

    import contextlib
    with (torch.cuda.stream(torch.cuda.Stream()) if a.side_stream else contextlib.nullcontext()):
        for i in range(3):
            o = step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            o = step(3 + i)
        host = (time.perf_counter() - t0) / a.steps      # the host's share: enqueueing the step's launches (no synchronisation inside)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
    fwd_flop = 1055342592.0 * (a.n / 256.0) ** 2 * a.batch * a.unroll
    line = {"what": "training step (hn_train_grad + hn_adam_step)", "n": a.n, "batch": a.batch, "unroll": a.unroll, "lanes": a.lanes, "ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": host * 1e3,
            "sample_iterations_per_s": a.batch * a.unroll / dt, "approx_tflops_fwd_plus_bwd": 3 * fwd_flop / dt / 1e12, "loss": float(o["loss"][0])}
    if a.cpu:
        from oracle import helmnet_oracle as O
        torch.set_num_threads(min(32, os.cpu_count() or 8))
        wts = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in s.f.state_dict().items()}
        t = O.SpectralTables(a.n, 8, 2, 1.0)
        args = [x.cpu() for x in (wf, res, st, k_sq, src)]
        t0 = time.perf_counter()
        loss, *_ = O.training_loss(args[0], args[1], args[2], args[3], args[4], wts, t, a.unroll)
        loss.backward()
        line["cpu_oracle_ms_per_step"] = (time.perf_counter() - t0) * 1e3
        line["cpu_threads"] = torch.get_num_threads()
        line["cpu_loss"] = float(loss)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
