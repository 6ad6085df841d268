"""Which level-0 kernels run in the eager loop, under capture (HN_OPT_GRAPH) and with two pipeline lanes -- run under rocprofv3 --kernel-trace --stats.

    rocprofv3 --kernel-trace --stats -d out -- python3 tools/pair_modes_check.py graph
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from helmnet_amd import IterativeSolver
    from helmnet_amd.phantoms import ring_sos_batch
    mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
    n, b = 256, 8
    sos = torch.from_numpy(ring_sos_batch(n, b, seed=1)).to("cuda:0")
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
    s.set_domain_size(n, source_location=[n // 8, n // 2])
    if mode == "graph":
        s.engine().set_option("graph", 1)
    elif mode == "lanes":
        s.engine().set_option("lanes", 2)
    o = s.forward(sos, num_iterations=40, residuals="norms")
    torch.cuda.synchronize()
    s.engine().check_async_errors()
    print(mode, "rmse", o["residual_norms"][-1].max().item())


if __name__ == "__main__":
    main()
