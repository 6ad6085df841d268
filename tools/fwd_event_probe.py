"""When does work that waits on the forward event of hn_train_grad actually run?  For a few (event flags, side-stream priority) variants: host time from
the launch of train_grad until (a) a tiny side-stream reduction + pinned copy behind the event has completed, (b) the whole step has.
Usage: python tools/fwd_event_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from helmnet_amd import IterativeSolver
    from helmnet_amd.phantoms import ring_sos_batch
    n, b = 96, 32
    s = IterativeSolver.from_exported_weights()
    s.to("cuda:0")
    s.set_domain_size(n, source_location=[n - 14, n // 2])
    tr = s.trainer()
    eng = s.engine()
    sos = torch.from_numpy(ring_sos_batch(n, b, seed=1)).to("cuda:0")
    out = s.forward(sos, num_iterations=3, return_wavefields=True, return_states=True)
    args = [out["wavefields"][-1].contiguous(), out["residuals"][-1].contiguous(), out["states"][-1].contiguous(),
            ((1.0 / sos) ** 2).contiguous(), s.source.detach().repeat(b, 1, 1, 1).contiguous()]
    blob = tr.weights
    host = torch.empty(b, dtype=torch.bool).pin_memory()
    for timing in (False, True):
        for prio in (0, -1):
            for what in ("reduce", "copy_only", "nothing"):
                fwd = torch.cuda.Event(enable_timing=timing)
                done = torch.cuda.Event()
                side = torch.cuda.Stream(device="cuda:0", priority=prio)
                eng.set_train_forward_event(fwd)
                ta, tb = [], []
                flag = torch.zeros(b, dtype=torch.bool, device="cuda:0")
                for _ in range(8):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    got = eng.train_grad(blob, *args, 10, 1e4)
                    t_enq = time.perf_counter() - t0
                    with torch.cuda.stream(side):
                        side.wait_event(fwd)
                        if what == "reduce":
                            host.copy_(got["residuals"][3].pow(2).mean((1, 2, 3)) < 1, non_blocking=True)
                        elif what == "copy_only":
                            host.copy_(flag, non_blocking=True)
                        done.record(side)
                    done.synchronize()
                    ta.append(time.perf_counter() - t0)
                    torch.cuda.synchronize()
                    tb.append(time.perf_counter() - t0)
                print(f"timing={timing} side priority={prio} work={what:9s}: enqueue {t_enq * 1e3:.2f} ms, side work done at {np.median(ta) * 1e3:.2f} ms, step done at {np.median(tb) * 1e3:.2f} ms")
    # the event alone, host-synchronised
    fwd = torch.cuda.Event()
    eng.set_train_forward_event(fwd)
    ta, tb = [], []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train_grad(blob, *args, 10, 1e4)
        fwd.synchronize()
        ta.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        tb.append(time.perf_counter() - t0)
    print(f"host waits on the forward event itself: at {np.median(ta) * 1e3:.2f} ms, step done at {np.median(tb) * 1e3:.2f} ms")
    eng.set_train_forward_event(None)


if __name__ == "__main__":
    main()
