"""End-to-end demonstration of the training half on the HIP library: a freshly initialised network (hybridnet.py:70-75) trained on synthetic
ring phantoms with the reference's recipe (replay buffer, 10 unrolled iterations, loss 1e4 * mean(res^2), value clipping, Adam(0.9, 0.95),
ReduceLROnPlateau), printing per epoch the mean training loss and a validation figure: the residual RMSE after max_iterations solver
iterations on held-out maps (validation_step, hybridnet.py:333-352).

    python tools/train_demo.py [--n 64] [--epochs 40] [--steps 25] [--batch 32]
"""
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
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--buffer", type=int, default=256)
    ap.add_argument("--max-iterations", type=int, default=100)
    a = ap.parse_args()
    from helmnet_amd import IterativeSolver
    from helmnet_amd.phantoms import ring_sos_batch
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    n = a.n
    s = IterativeSolver(domain_size=n, k=1.0, omega=1, PMLsize=8, sigma_max=2, source_location=[n - 14, n // 2], activation_function="prelu",
                        batch_size=a.batch, buffer_size=a.buffer, learning_rate=1e-3, minimum_learning_rate=1e-4, weight_decay=1e-6,
                        gradient_clip_val=1, max_iterations=a.max_iterations, source_amplitude=10, unrolling_steps=10).to("cuda:0")
    sos_train = torch.from_numpy(ring_sos_batch(n, max(a.buffer, 512), seed=100))
    sos_val = torch.from_numpy(ring_sos_batch(n, 16, seed=999)).to("cuda:0")
    tr = s.trainer()
    tr.fill_replay_buffer(sos_train)

    def validate():
        """RMSE of the residual after max_iterations free-running iterations on held-out maps: with the training source (test_step,
        hybridnet.py:299-314) and with one random source per sample (validation_step, :333-352)."""
        np.random.seed(1234)   # the same random source locations every time
        v = s.validation_step(sos_val)
        t = s.test_step(sos_val)   # resets the source to hparams.source_location
        fixed = float(t["losses"][:, -1].pow(2).mean().sqrt())
        return {"val_rmse_training_source": fixed, "val_rmse_random_sources": float(v["loss"])}

    log = [{"epoch": 0, **validate(), "lr": tr.lr}]
    print(json.dumps(log[-1]), flush=True)
    t0 = time.perf_counter()
    for epoch in range(1, a.epochs + 1):
        for i in range(a.steps):
            idx = np.random.choice(len(sos_train), a.batch, replace=False)
            tr.training_step(sos_train[idx].to("cuda:0"), i)
        mean = tr.training_epoch_end()
        if epoch % 5 == 0 or epoch == a.epochs:
            log.append({"epoch": epoch, "train_loss_mean": mean, **validate(), "lr": tr.lr, "new_sos": tr.new_sos,
                        "maxiter": min(tr.current_epoch * 20 + 1, a.max_iterations), "seconds": round(time.perf_counter() - t0, 1)})
            print(json.dumps(log[-1]), flush=True)
    torch.cuda.synchronize()
    print(json.dumps({"steps": a.epochs * a.steps, "seconds": round(time.perf_counter() - t0, 1),
                      "val_rmse_training_source_start": log[0]["val_rmse_training_source"], "val_rmse_training_source_end": log[-1]["val_rmse_training_source"]}))


if __name__ == "__main__":
    main()
