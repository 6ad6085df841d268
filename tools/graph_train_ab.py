"""Does replaying the training step as ONE captured HIP graph beat launching its ~620 kernels?  (The step is a chain of small dependent
launches)  The step is captured with torch.cuda.graph on a side stream -- hn_train_grad only enqueues work on the caller's stream --
and replayed; same inputs every time, so the job tables the kernels read stay valid.  Usage: python tools/graph_train_ab.py [--batch 32]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    from helmnet_amd import IterativeSolver
    from helmnet_amd.engine import pack_weights
    from helmnet_amd.phantoms import ring_sos_batch
    dev = "cuda:0"
    s = IterativeSolver.from_exported_weights()
    s.to(dev)
    s.set_domain_size(a.n, source_location=[a.n - 14, a.n // 2])
    eng = s.engine()
    sos = torch.from_numpy(ring_sos_batch(a.n, a.batch, seed=5)).to(dev)
    out = s.forward(sos, num_iterations=5, return_wavefields=True, return_states=True)
    wf, res, st = out["wavefields"][-1].contiguous(), out["residuals"][-1].contiguous(), out["states"][-1].contiguous()
    k_sq = ((1.0 / sos) ** 2).contiguous()
    src = s.source.detach().repeat(a.batch, 1, 1, 1).contiguous()
    w = torch.from_numpy(pack_weights(dict(s.f.state_dict()))).to(dev)
    g = torch.zeros_like(w)
    T = 10
    hist = [torch.empty(T, *wf.shape, device=dev), torch.empty(T, *res.shape, device=dev), torch.empty(T, *st.shape, device=dev)]
    loss = torch.zeros(1, device=dev)

    def step():
        return eng.train_grad(w, wf, res, st, k_sq, src, T, 1e4, grad=g)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            o = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            o = step()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / a.steps
    ref = g.clone()
    line = {"n": a.n, "batch": a.batch, "eager_ms": round(eager * 1e3, 3)}
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            o = step()
        torch.cuda.synchronize()
        g.zero_()
        graph.replay()
        torch.cuda.synchronize()
        line["graph_gradient_equals_eager"] = bool(torch.equal(g, ref))
        t0 = time.perf_counter()
        for _ in range(a.steps):
            graph.replay()
        torch.cuda.synchronize()
        line["graph_replay_ms"] = round((time.perf_counter() - t0) / a.steps * 1e3, 3)
    except Exception as e:   # capture is an experiment: report, do not fail
        line["graph_error"] = repr(e)[:300]
    print(json.dumps(line))


if __name__ == "__main__":
    main()
