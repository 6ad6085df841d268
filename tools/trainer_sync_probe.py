"""Which host<->device synchronisations does one ``Trainer.training_step`` contain?  Runs a few steps under torch's sync debug mode ("warn") and
prints every warning with the Python line that caused it, then times the host side of a step (time until training_step returns) next to the
device side.  Usage: python tools/trainer_sync_probe.py [--n 96] [--batch 32]"""
import argparse
import os
import random
import sys
import time
import traceback
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    from helmnet_amd import IterativeSolver
    from helmnet_amd.phantoms import ring_sos_batch
    torch.manual_seed(0); np.random.seed(0); random.seed(0)
    s = IterativeSolver.from_exported_weights()
    s.to("cuda:0")
    s.hparams.batch_size, s.hparams.buffer_size = a.batch, 256
    s.set_domain_size(a.n, source_location=[a.n - 14, a.n // 2])
    sos_train = torch.from_numpy(ring_sos_batch(a.n, 512, seed=100))
    tr = s.trainer()
    tr.current_epoch = 10
    tr.fill_replay_buffer(sos_train)
    batches = [sos_train[np.random.choice(len(sos_train), a.batch, replace=False)].to("cuda:0") for _ in range(40)]
    for i in range(4):
        tr.training_step(batches[i], i)
    torch.cuda.synchronize()

    def show(message, category, filename, lineno, file=None, line=None):
        here = [f for f in traceback.extract_stack() if "helmnet_amd" in f.filename]
        print("SYNC:", str(message).split("\n")[0], "<-", ", ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in here[-3:]))
    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    for i in range(2):
        print(f"--- step {i}")
        tr.training_step(batches[4 + i], i)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    # host time of a step (until training_step returns) vs wall time per step
    host = []
    t0 = time.perf_counter()
    for i in range(30):
        h0 = time.perf_counter()
        tr.training_step(batches[6 + i], i)
        host.append(time.perf_counter() - h0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 30
    print(f"host side of a step: median {np.median(host) * 1e3:.2f} ms; wall per step {wall * 1e3:.2f} ms")
    # where the host time goes
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20):
        tr.training_step(batches[i], i)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
