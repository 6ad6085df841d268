"""Where do the ~0.4 ms go that Trainer.training_step takes beyond hn_train_grad + hn_adam_step?  The step written out with switches (measurement only:
variants that skip parts of the host logic are NOT training steps).  Usage: python tools/trainer_step_ablation.py [--batch 32]"""
import argparse
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=60)
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
    batches = [sos_train[np.random.choice(len(sos_train), a.batch, replace=False)].to("cuda:0") for _ in range(8)]
    rb = tr.replaybuffer
    side, mask_done = torch.cuda.Stream(device="cuda:0"), torch.cuda.Event()
    keep_host = torch.empty(a.batch, dtype=torch.bool).pin_memory()

    def step(sos_batch, mode):
        hp = s.hparams
        maxiter = min([tr.current_epoch * 20 + 1, hp.max_iterations])
        if "fixed_sample" in mode:
            sample = step.cache
        else:
            sample = rb.sample(hp.batch_size)
        wavefields, h_states, k_sqs, residual, sources, timesteps, indices = sample
        out = tr.loss_and_grad(wavefields, h_states, k_sqs, residual, sources)
        T = out["residuals"].shape[0]
        iteration = np.random.choice(T)
        res_it, wf_it, st_it = out["residuals"][iteration], out["wavefields"][iteration], out["states"][iteration]
        nb = wavefields.shape[0]
        if "no_mask" in mode:
            tr.optimizer_step()
            keep = np.ones(nb, dtype=bool)
            keep[:3] = False
        elif "main_stream_mask" in mode:      # torch's reduction behind the backward pass on the main stream, blocking read (rounds 3 / early 4)
            tr.optimizer_step()
            keep = (res_it.pow(2).mean((1, 2, 3)) < 1).cpu().numpy()
        elif "side_stream_mask" in mode:      # torch's reduction on a second stream behind the forward event
            with torch.cuda.stream(side):
                side.wait_event(tr._fwd_event)
                keep_host[:nb].copy_(res_it.pow(2).mean((1, 2, 3)) < 1, non_blocking=True)
                mask_done.record(side)
            tr.optimizer_step()
            mask_done.synchronize()
            keep = keep_host[:nb].numpy().copy()
        else:                                 # the product: the forward sweep's own sums, copied to pinned memory by the library before the forward event
            tr.optimizer_step()
            tr._fwd_event.synchronize()
            keep = tr._sumsq_host[: T * nb].view(T, nb)[iteration].numpy() / np.float32(res_it[0].numel()) < 1
        new_timesteps = np.asarray(timesteps, dtype=np.int64) + iteration + 1
        keep = keep & (new_timesteps < maxiter)
        fresh = np.nonzero(~keep)[0]
        if "no_update" not in mode:
            rb.update(indices, np.where(keep, new_timesteps, 0), wavefield=wf_it, hidden_state=st_it, residual=res_it)
        if fresh.size and "no_fresh" not in mode:
            maps = torch.stack([random.choice(sos_batch) for _ in fresh])
            k_sq, res, src = tr._fresh_fields(maps, zeros=False)
            rb.update(indices[fresh], np.zeros(fresh.size, dtype=np.int64), zero=("wavefield", "hidden_state"), k_sq=k_sq, residual=res, source=src)
        return fresh.size

    step.cache = rb.sample(a.batch)
    modes = ["full", "no_fresh", "no_fresh no_update", "no_fresh no_update fixed_sample", "no_mask", "no_mask no_fresh no_update fixed_sample", "main_stream_mask", "side_stream_mask"]
    for rep in range(2):
        for mode in modes:
            for i in range(5):
                step(batches[i % 8], mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nf = 0
            for i in range(a.steps):
                nf += step(batches[i % 8], mode)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.steps
            print(f"{mode:45s} {dt * 1e3:7.3f} ms / step   fresh/step {nf / a.steps:.2f}", flush=True)
        # bare library calls
        wavefields, h_states, k_sqs, residual, sources, _, _ = step.cache
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            tr.loss_and_grad(wavefields, h_states, k_sqs, residual, sources); tr.optimizer_step()
        torch.cuda.synchronize()
        print(f"{'library calls alone':45s} {(time.perf_counter() - t0) / a.steps * 1e3:7.3f} ms / step", flush=True)


if __name__ == "__main__":
    main()
