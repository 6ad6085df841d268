"""Multi-GPU use of the solver: one process per GPU, batches of SoS maps sharded across ranks.

Samples never interact in the reference loop (helmnet/hybridnet.py:654-697: every op is
per-sample; source / sigmas / weights are broadcast constants), so the path shards by batch
slices with NO data-path collective.  The only exchange is the tiny residual-norm all-reduce
used for a global convergence check (per-sample RMSE, hybridnet.py:295-297) and an optional
final gather of the wavefields.  Backend "nccl" is RCCL on ROCm; the same code runs on "gloo"
for the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def _ready() -> bool:
    return dist.is_available() and dist.is_initialized()


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a batch of ``total`` samples owned by ``rank``; sizes differ
    by at most one and concatenating the slices in rank order restores the batch."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return batch[lo:hi]


def allreduce_residual_norms(norms: torch.Tensor, op: str = "max") -> torch.Tensor:
    """Reduce per-sample residual RMSEs over all ranks.

    op = "max": [1] worst RMSE anywhere (all-converged test);  "sum"/"mean": [1] sum / mean of
    the per-sample RMSEs over the global batch.  Without an initialised process group this is
    the local reduction, so single-GPU code paths are identical.
    """
    if op not in ("max", "sum", "mean"):
        raise ValueError("op must be 'max', 'sum' or 'mean'")
    flat = norms.reshape(-1).float()
    if op == "max":
        out = flat.max().reshape(1) if flat.numel() else flat.new_zeros(1)
        if _ready():
            dist.all_reduce(out, op=dist.ReduceOp.MAX)
        return out
    acc = torch.stack([flat.sum(), flat.new_tensor(float(flat.numel()))])
    if _ready():
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return (acc[0] if op == "sum" else acc[0] / acc[1].clamp_min(1)).reshape(1)


def gather_batch(local: torch.Tensor, total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Collect the per-rank result slices on ``dst`` in rank order (ragged last shards allowed)."""
    if not _ready():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = local.new_zeros((width,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


def solve_sharded(solve: Callable[[torch.Tensor, int], dict], sos_maps: torch.Tensor, num_iterations: int,
                  tol: Optional[float] = None, check_every: int = 50, gather: bool = False) -> dict:
    """Run ``solve(local_sos, n_iter)`` on this rank's shard of ``sos_maps``.

    ``solve`` continues from its own state when called again (e.g. a closure over
    IterativeSolver.forward for the first chunk and n_steps afterwards) and returns a dict with
    "wavefield" [b,2,N,N] and "rmse" [b].  With ``tol`` the loop stops once the WORST RMSE over
    all ranks drops below it, checked every ``check_every`` iterations with one all-reduce.
    """
    total = sos_maps.shape[0]
    if _ready() and total < dist.get_world_size():
        # every rank evaluates the same condition, so every rank raises: no rank is left waiting in a collective
        raise ValueError(f"{total} maps cannot be sharded over {dist.get_world_size()} ranks (an empty shard has nothing to solve)")
    local = shard_batch(sos_maps) if _ready() else sos_maps
    done, out, worst = 0, None, None
    while done < num_iterations:
        chunk = min(check_every if tol is not None else num_iterations, num_iterations - done)
        out = solve(local, chunk)
        done += chunk
        worst = allreduce_residual_norms(out["rmse"], "max")
        if tol is not None and worst.item() < tol:
            break
    result = {"wavefield": out["wavefield"], "rmse": out["rmse"], "iterations": done, "worst_rmse": worst}
    if gather:
        result["wavefield_all"] = gather_batch(out["wavefield"], total)
    return result
