"""End-to-end time of ``Trainer.training_step`` (replay-buffer sample -> hn_train_grad -> hn_adam_step -> buffer refill, the host logic of
hybridnet.py:385-505) next to the time of the two library calls alone.  Usage: python tools/bench_trainer_loop.py [--n 96] [--steps 40]"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--buffer", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    from helmnet_amd import IterativeSolver
    from helmnet_amd.phantoms import ring_sos_batch
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    s = IterativeSolver.from_exported_weights()
    s.to("cuda:0")
    s.hparams.batch_size, s.hparams.buffer_size = a.batch, max(a.buffer, 2 * a.batch)
    s.set_domain_size(a.n, source_location=[a.n - 14, a.n // 2])
    sos_train = torch.from_numpy(ring_sos_batch(a.n, max(a.buffer, 2 * a.batch, 512), seed=100))
    tr = s.trainer()
    tr.current_epoch = 10     # maxiter = 201: most slots are advanced, a few are re-drawn (steady state of a run)
    t0 = time.perf_counter()
    tr.fill_replay_buffer(sos_train)
    torch.cuda.synchronize()
    fill = time.perf_counter() - t0
    batches = [sos_train[np.random.choice(len(sos_train), a.batch, replace=False)].to("cuda:0") for _ in range(a.steps + 5)]
    for i in range(5):
        tr.training_step(batches[i], i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    new = 0
    for i in range(a.steps):
        new += tr.training_step(batches[5 + i], i)["new_sos"]
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / a.steps
    # the two library calls alone on one sampled batch
    wavefields, h_states, k_sqs, residual, sources, _, _ = tr.replaybuffer.sample(a.batch)
    for _ in range(3):
        tr.loss_and_grad(wavefields, h_states, k_sqs, residual, sources); tr.optimizer_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.loss_and_grad(wavefields, h_states, k_sqs, residual, sources); tr.optimizer_step()
    torch.cuda.synchronize()
    lib = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"n": a.n, "batch": a.batch, "buffer": a.buffer, "training_step_ms": round(total * 1e3, 3), "library_calls_ms": round(lib * 1e3, 3),
                      "host_logic_ms": round((total - lib) * 1e3, 3), "fresh_maps_per_step": new / a.steps, "fill_replay_buffer_s": round(fill, 3)}))


if __name__ == "__main__":
    main()
