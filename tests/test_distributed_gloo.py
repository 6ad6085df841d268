"""The N > 1 path on CPU: world_size-2 gloo processes exercise the sharding, the residual-norm
all-reduce and the gather exactly as the GPU path uses them (the per-shard solve is the CPU
oracle here -- test infrastructure standing in for the HIP engine)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helmnet_amd.distributed import (allreduce_residual_norms, gather_batch, shard_batch, shard_bounds,
                                             solve_sharded)
        from helmnet_amd.phantoms import ring_sos_batch
        from oracle import helmnet_oracle as O
        with np.load(os.path.join(os.path.dirname(__file__), "golden", "jcp_weights.npz")) as z:
            w = {k: torch.from_numpy(z[k]) for k in z.files}
        n, total = 32, 5                                   # ragged: shards of 3 and 2
        sos = torch.from_numpy(ring_sos_batch(n, total, seed=1))
        t = O.SpectralTables(n, 8, 2, 1.0)
        src = O.point_source_map(n, [8, 16], 10.0)
        mine = shard_batch(sos)
        assert mine.shape[0] == (3 if rank == 0 else 2)
        state = {}

        def solve(local, iters):
            if not state:
                k_sq, wf = O.get_initials(local, 1.0)
                state.update(k_sq=k_sq, wf=wf, st=[torch.zeros(local.shape[0], 2, s, s) for s in O.state_dims(n, 4)])
                state["res"] = O.get_residual(wf, k_sq, src, t)
            for _ in range(iters):
                state["wf"], state["res"], state["st"] = O.single_step(state["wf"], state["k_sq"], state["res"],
                                                                       state["st"], w, src, t)
            return {"wavefield": state["wf"], "rmse": O.test_loss_function(state["res"])}

        out = solve_sharded(solve, sos, 12, tol=None, gather=True)
        full = O.solve(sos, w, src, t, 12)
        lo, hi = shard_bounds(total, rank, world)
        # sharded == unsharded for the same samples.  (Bitwise on the HIP path -- see
        # test_gpu_parity.py::test_batch_samples_are_independent_and_deterministic; the CPU stand-in's
        # oneDNN/MKL kernels pick batch-size dependent blockings, hence the round-off tolerance.)
        assert (out["wavefield"] - full["wavefield"][lo:hi]).abs().max() <= 2e-5
        want_worst = O.test_loss_function(full["residual"]).max()
        assert torch.allclose(out["worst_rmse"], want_worst.reshape(1), rtol=1e-3)
        if rank == 0:
            assert (out["wavefield_all"] - full["wavefield"]).abs().max() <= 2e-5
            assert out["wavefield_all"].shape == full["wavefield"].shape
        else:
            assert out["wavefield_all"] is None
        m = allreduce_residual_norms(out["rmse"], "mean")
        assert torch.allclose(m, O.test_loss_function(full["residual"]).mean().reshape(1), rtol=1e-3)
        # early stop is decided globally: every rank stops at the same iteration
        state.clear()
        out2 = solve_sharded(solve, sos, 40, tol=float(want_worst) * 1.5, check_every=4)
        its = torch.tensor([out2["iterations"]])
        both = [torch.zeros_like(its) for _ in range(world)]
        dist.all_gather(both, its)
        assert both[0] == both[1] and out2["iterations"] <= 12
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_solve():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_shard_bounds_cover_batch():
    from helmnet_amd.distributed import allreduce_residual_norms, shard_bounds
    for total in (1, 5, 32, 256):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(total, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)
    x = torch.tensor([1.0, 3.0, 2.0])
    assert allreduce_residual_norms(x, "max").item() == 3.0      # no process group: local reduction
    assert allreduce_residual_norms(x, "mean").item() == 2.0


def test_bench_starts_its_own_ranks_and_says_what_it_needs():
    """`python bench.py --gpus 2` without a launcher spawns two child ranks before any GPU call; on a box with fewer than two devices every
    rank that has no device says so and the launcher exits non-zero (no hang in the rendezvous, no 'use torch.distributed.run' hint)."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the launcher would run the benchmark")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 devices on this node" in r.stderr and "torch.distributed.run" not in r.stderr, r.stderr[-2000:]
    assert "rank 1" in r.stderr      # the child without a device reported it itself
