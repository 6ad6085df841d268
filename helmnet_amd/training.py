"""Training half of the reference's ``IterativeSolver`` on the HIP library.

Mirrors, without Lightning: ``helmnet/replaybuffer.py:1-47`` (``Experience`` / ``ReplayBuffer``), the buffer fill of
``train_dataloader`` (hybridnet.py:192-218), ``training_step`` (:385-505), ``on_after_backward`` (:172-176),
``configure_optimizers`` (:250-283: Adam(0.9, 0.95) + ReduceLROnPlateau on the epoch-mean training loss) and
``training_epoch_end`` (:379-383).  The arithmetic -- ten unrolled solver iterations, the loss, back-propagation through
time, gradient clipping and the Adam update -- runs in libhelmnet_hip.so (``hn_train_grad`` / ``hn_adam_step``); this
module holds the flat parameter / gradient / moment tensors, the replay buffer and the host-side bookkeeping.

Data-parallel training (the reference trains with Lightning DDP, train.py:103-112): one process per GPU.  The replicas start
from rank 0's parameters and Adam moments (``broadcast_from_rank0`` in ``Trainer.__init__`` / ``load_state_dict``: what DDP does
when it wraps the module), every rank draws its own replay-buffer batch, the flat gradient (one 193 KB bucket) is averaged
with a single all-reduce over RCCL between ``hn_train_grad`` and ``hn_adam_step`` (``allreduce_gradients``), and the
epoch-mean loss the learning-rate scheduler monitors is averaged over the ranks (``allreduce_mean_scalar``): same weights,
same moments, same gradient, same learning rate on every rank, so the replicas stay bit-identical.
"""
from __future__ import annotations

import collections
from random import choice
from typing import List, Optional, Sequence

import numpy as np
import torch

from .engine import pack_weights, unpack_weights, weight_shapes

# replaybuffer.py:8-18
Experience = collections.namedtuple(
    "Experience", field_names=["wavefield", "hidden_state", "k_sq", "residual", "source", "iteration"]
)


def _index_on(device, idx_host) -> torch.Tensor:
    """A host index array as an int64 tensor on ``device`` -- through pinned memory and a non-blocking copy on a GPU: a copy from pageable
    memory would make the host wait for everything already enqueued on the stream (the backward pass of the step in flight)."""
    t = torch.from_numpy(np.ascontiguousarray(idx_host, dtype=np.int64))
    return t.pin_memory().to(device, non_blocking=True) if torch.device(device).type == "cuda" else t.to(device)


class ReplayBuffer:
    """The reference's replay buffer (replaybuffer.py:20-47: ``capacity`` slots addressed by index, ``sample`` draws distinct
    slots with ``np.random.choice(capacity, batch_size, replace=False)``) kept as ONE pre-allocated ``[capacity, ...]`` tensor per
    field on the device of the first experience written: ``sample`` is an ``index_select`` per field and a training step writes
    its advanced / fresh experiences back with one ``index_copy_`` per field (``write``) instead of ``batch_size`` Python-side
    clones and a ``torch.stack`` of 5 x ``batch_size`` tensors.  Under a seed the sampled indices are the reference's."""

    FIELDS = ("wavefield", "hidden_state", "k_sq", "residual", "source")

    def __init__(self, capacity: int, engine=None):
        self.capacity = capacity
        self.engine = engine      # helmnet_amd.engine.Engine: rows move with hn_rows_gather / hn_rows_scatter (one launch per call, the slot list in the
                                  # kernel arguments); None (host tensors, tests): the same with torch's index_select / index_copy_
        self.fields: Optional[dict] = None                       # name -> [capacity, ...] tensor
        self.iteration = np.zeros(capacity, dtype=np.int64)      # nominal solver iteration of every slot (host side: it only feeds Python logic)
        self.filled = np.zeros(capacity, dtype=bool)
        self.last_sample_index: Optional[torch.Tensor] = None

    def __len__(self):
        return self.capacity

    def _ensure(self, like: dict):
        if self.fields is None:
            self.fields = {k: torch.empty((self.capacity,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device) for k, v in like.items()}

    def write(self, indices, wavefield, hidden_state, k_sq, residual, source, iterations):
        """Batched ``append``: slot ``indices[j]`` <- the j-th row of every field (``indices``: 1-D integer tensor / array)."""
        vals = dict(zip(self.FIELDS, (wavefield, hidden_state, k_sq, residual, source)))
        self._ensure(vals)
        idx_host = np.asarray(indices.cpu() if torch.is_tensor(indices) else indices, dtype=np.int64).reshape(-1)
        if idx_host.size == 0:
            return
        if self._native():
            self.engine.rows_scatter([self.fields[k] for k in self.FIELDS], idx_host, [self._row(k, v) for k, v in vals.items()])
        else:
            idx = _index_on(next(iter(self.fields.values())).device, idx_host)
            for k, v in vals.items():
                buf = self.fields[k]
                buf.index_copy_(0, idx, v.to(device=buf.device, dtype=buf.dtype))
        self.iteration[idx_host] = np.asarray(iterations, dtype=np.int64).reshape(-1)
        self.filled[idx_host] = True

    def _native(self) -> bool:
        return self.engine is not None and self.fields is not None and all(f.device == self.engine.device and f.dtype == torch.float32 for f in self.fields.values())

    def _row(self, k, v):
        buf = self.fields[k]
        return v.to(device=buf.device, dtype=buf.dtype)

    def update(self, indices, iterations, index_device: Optional[torch.Tensor] = None, zero: Sequence[str] = (), **fields):
        """Rewrite SOME fields of already written slots: slot ``indices[j]`` <- row j of every tensor in ``fields`` (one ``index_copy_`` each), zeros for the
        fields named in ``zero`` (one ``index_fill_`` each), a tensor with ONE row goes to every slot; the other fields keep what they hold.  ``index_device``: ``indices`` on the device, if the caller has it."""
        idx_host = np.asarray(indices, dtype=np.int64).reshape(-1)
        if idx_host.size == 0:
            return
        if not self.filled[idx_host].all():
            raise ValueError("update of an empty replay-buffer slot")
        if self._native():
            names = list(fields) + list(zero)
            self.engine.rows_scatter([self.fields[k] for k in names], idx_host, [self._row(k, v) for k, v in fields.items()] + [None] * len(zero))
        else:
            idx = index_device if index_device is not None else _index_on(next(iter(self.fields.values())).device, idx_host)
            for k, v in fields.items():
                buf = self.fields[k]
                v = v.expand(idx_host.size, *v.shape[1:]) if v.shape[0] == 1 else v        # one row for every slot
                buf.index_copy_(0, idx, v.to(device=buf.device, dtype=buf.dtype))
            for k in zero:
                self.fields[k].index_fill_(0, idx, 0)
        self.iteration[idx_host] = np.asarray(iterations, dtype=np.int64).reshape(-1)

    def append(self, experience, index):   # replaybuffer.py:29-30
        self.write([index], *(f.unsqueeze(0) for f in experience[:5]), [experience[5]])

    @property
    def buffer(self):
        """The reference's list view: an ``Experience`` of views per written slot, ``None`` for an empty one."""
        return [Experience(*(self.fields[k][i] for k in self.FIELDS), int(self.iteration[i])) if self.filled[i] else None for i in range(self.capacity)]

    def sample(self, batch_size: int):     # replaybuffer.py:32-47
        if batch_size > self.capacity:
            batch_size = self.capacity
        indices = np.random.choice(self.capacity, batch_size, replace=False)
        if not self.filled[indices].all():
            raise ValueError("sampled an empty replay-buffer slot: fill the buffer first (fill_replay_buffer)")
        if self._native():
            out = self.engine.rows_gather([self.fields[k] for k in self.FIELDS], indices)
        else:
            idx = _index_on(next(iter(self.fields.values())).device, indices)
            self.last_sample_index = idx            # (on the device already: ``update`` of the same slots needs no second copy)
            out = [self.fields[k].index_select(0, idx) for k in self.FIELDS]
        return (*out, tuple(int(i) for i in self.iteration[indices]), indices)


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_from_rank0(*tensors: torch.Tensor) -> None:
    """What wrapping a module in DDP does once (train.py:103-112 under Lightning): every rank starts from rank 0's parameters (here also
    its optimiser moments, for a resumed run).  A no-op without an initialised process group."""
    if _world_size() > 1:
        import torch.distributed as dist
        for t in tensors:
            dist.broadcast(t, src=0)


def allreduce_mean_scalar(value: float, device="cpu") -> float:
    """Mean of a per-rank scalar over the ranks (the epoch-mean training loss ReduceLROnPlateau monitors: Lightning logs it with the
    mean over ranks, so every replica's scheduler sees the same number and the learning rates cannot drift apart)."""
    if _world_size() > 1:
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0]) / dist.get_world_size()
    return value


def allreduce_gradients(grad: torch.Tensor) -> torch.Tensor:
    """Average the flat gradient over the ranks of the default process group (DDP's gradient all-reduce as ONE bucket);
    a no-op without an initialised group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM)
        grad /= dist.get_world_size()
    return grad


def trainable_mask(depth: int, activation: str, state_depth: Optional[int] = None) -> np.ndarray:
    """1 for every blob entry the optimiser may change.  The slope slot of a DoubleConv is a parameter only for PReLU
    (architectures.py:32-33); for the parameter-free activations it holds a constant.  An encoder level WITHOUT state
    (d >= state_depth, architectures.py:353) lives in the blob as its zero-padded stateful equivalent (engine.pack_weights): the
    two padded input channels of its conv_signal and its whole (all-zero) conv_state are not parameters of the module."""
    state_depth = depth if state_depth is None else state_depth
    parts = []
    for name, shape in weight_shapes(depth).items():
        m = np.ones(shape, np.uint8)
        if name.endswith(".double_conv.1.weight") and activation.lower() != "prelu":
            m[...] = 0
        if name.startswith("enc."):
            d = int(name.split(".")[1])
            if d >= state_depth:
                if ".conv_state." in name:
                    m[...] = 0
                elif name.endswith(".conv_signal.double_conv.0.weight"):
                    m[:, 8:] = 0
        parts.append(m.reshape(-1))
    return np.concatenate(parts)


class Trainer:
    """Optimiser state + training loop for one ``IterativeSolver`` (which must live on the GPU).

    ``solver.f``'s parameters stay the module-level copy of the weights (``state_dict`` / checkpoints / inference);
    the trainer works on a flat device blob and writes it back with ``sync_to_module`` (done by ``training_epoch_end``
    and on demand)."""

    def __init__(self, solver, grad_reduce=allreduce_gradients):
        hp = solver.hparams
        if hp.optimizer.lower() != "adam":
            raise NotImplementedError("The optimizer {} is not implemented".format(hp.optimizer))  # hybridnet.py:259-262
        if hp.loss != "mse":
            raise NotImplementedError("The loss function {} is not implemented".format(hp.loss))   # :291-294
        if hp.minimum_learning_rate > hp.learning_rate:                                             # :264-269
            raise ValueError("Minimum learning rate ({}) must be smaller than the starting learning rate ({})".format(
                hp.minimum_learning_rate, hp.learning_rate))
        self.solver = solver
        self.grad_reduce = grad_reduce
        self.engine = solver.engine()
        dev = self.engine.device
        f = solver.f
        self.depth, self.activation, self.state_depth = f.depth, f.activation_function, f.state_depth
        blob = pack_weights(dict(f.state_dict()), f.depth, f.activation_function, f.state_depth)
        self.weights = torch.from_numpy(blob).to(dev)
        self.grad = torch.zeros_like(self.weights)
        self.exp_avg = torch.zeros_like(self.weights)
        self.exp_avg_sq = torch.zeros_like(self.weights)
        self.trainable = torch.from_numpy(trainable_mask(f.depth, f.activation_function, f.state_depth)).to(dev)
        broadcast_from_rank0(self.weights, self.exp_avg, self.exp_avg_sq)   # DDP: every replica starts from rank 0's state
        self.step_count = 0
        self.current_epoch = 0
        self.global_step = 0
        self.lr = float(hp.learning_rate)
        # ReduceLROnPlateau(mode="min", factor=0.5, patience=10, min_lr) monitoring the epoch-mean training loss (:270-283);
        # torch's scheduler does the bookkeeping on a one-parameter stand-in optimiser whose lr is read back
        self._lr_holder = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=self.lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self._lr_holder, mode="min", factor=0.5, patience=10,
                                                                     min_lr=float(hp.minimum_learning_rate))
        self.replaybuffer = ReplayBuffer(hp.buffer_size, engine=self.engine)
        self.epoch_losses: List[torch.Tensor] = []
        self.new_sos = 0
        # The refill decision of training_step needs ONE number per sample on the host (is the advanced residual still bounded?).  It depends on the
        # forward sweep only, and the forward sweep already sums res^2 per (iteration, sample) for the loss: the library copies that table to pinned
        # memory and records `_fwd_event` behind the forward sweep, so the host waits for that instead of draining the backward pass too.
        self._fwd_event = torch.cuda.Event()
        self._sumsq_host = None
        self._reserve_sumsq(int(hp.batch_size) * int(hp.unrolling_steps))

    # ------------------------------------------------------------------ weights -----------------
    def sync_to_module(self):
        """Write the trained blob back into ``solver.f``'s parameters (so that inference, ``state_dict`` and checkpoints see it)."""
        sd = unpack_weights(self.weights, self.depth)
        with torch.no_grad():
            for name, p in self.solver.f.named_parameters():
                v = torch.from_numpy(sd[name])
                if v.dim() == 4 and v.shape[1] != p.shape[1]:     # conv_signal of a level without state: drop the zero-padded channels
                    v = v[:, : p.shape[1]]
                p.copy_(v.reshape(p.shape))

    def state_dict(self) -> dict:
        """Everything needed to resume: parameters, Adam moments and step, learning rate and scheduler state."""
        return {"weights": self.weights.cpu(), "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "step": self.step_count,
                "lr": self.lr, "epoch": self.current_epoch, "global_step": self.global_step, "scheduler": self.scheduler.state_dict()}

    def load_state_dict(self, sd: dict):
        for k in ("weights", "exp_avg", "exp_avg_sq"):
            getattr(self, k).copy_(sd[k].to(self.weights.device))
        self.step_count, self.lr, self.current_epoch, self.global_step = int(sd["step"]), float(sd["lr"]), int(sd["epoch"]), int(sd["global_step"])
        sched = sd["scheduler"]
        if _world_size() > 1:   # a collective: EVERY rank must call load_state_dict.  Rank 0's file wins for everything the replicas must agree on -- not only
            import torch.distributed as dist   # the tensors: the Adam step (bias correction), the learning rate, the epoch and the plateau scheduler's state
            box = [(self.step_count, self.lr, self.current_epoch, self.global_step, sched)]
            dist.broadcast_object_list(box, src=0)
            self.step_count, self.lr, self.current_epoch, self.global_step, sched = box[0]
        self.scheduler.load_state_dict(sched)
        self._lr_holder.param_groups[0]["lr"] = self.lr
        broadcast_from_rank0(self.weights, self.exp_avg, self.exp_avg_sq)
        self.sync_to_module()

    # ------------------------------------------------------------------ replay buffer -----------
    def _fresh_fields(self, sos_maps: torch.Tensor, zeros: bool = True):
        """Initial (zero wavefield) experiences of a batch of [n, 1, N, N] sound-speed maps (hybridnet.py:203-216, 454-462) as the five
        field tensors: ONE get_initials and ONE residual call for the whole batch.  ``zeros=False``: only (k_sq, residual, source) -- the
        caller zero-fills the wavefield and hidden-state rows in place."""
        s = self.solver
        s.reset_source()
        k_sq, wf = s.get_initials(sos_maps.float().to(s.device))
        # after set_source_maps(sources) the reference's self.source holds one map per sample and its get_residual broadcasts
        # (:556); the maps of a batch are equal unless the caller set different ones -- the first one is the source of a fresh slot
        src0 = s.source[:1].detach().float().contiguous()
        res = s.engine().residual(wf, k_sq.contiguous(), src0)
        if not zeros:
            return k_sq, res, src0          # (one source row for every slot)
        src = src0.expand(wf.shape[0], -1, -1, -1)
        s.f.clear_states(wf)
        h = s.f.get_states(flatten=True)
        return wf, h, k_sq, res, src

    def _fresh_experience(self, sos_map: torch.Tensor, iteration: int) -> Experience:
        return Experience(*(f[0] for f in self._fresh_fields(sos_map)), iteration)

    def fill_replay_buffer(self, sos_train, chunk: int = 64):
        """hybridnet.py:192-218: one fresh experience per slot, slot ``c`` starting at nominal iteration ``10 c``."""
        with torch.no_grad():
            cap = len(self.replaybuffer)
            for c0 in range(0, cap, chunk):
                idx = list(range(c0, min(c0 + chunk, cap)))
                maps = torch.stack([torch.as_tensor(sos_train[c]) for c in idx])
                self.replaybuffer.write(idx, *self._fresh_fields(maps), [10 * c for c in idx])

    # ------------------------------------------------------------------ one step ----------------
    def _reserve_sumsq(self, floats: int):
        if self._sumsq_host is None or self._sumsq_host.numel() < floats:
            self._sumsq_host = torch.zeros(max(int(floats), 1 << 16), dtype=torch.float32).pin_memory()
        if getattr(self.engine, "_fwd_sumsq", None) is not self._sumsq_host:      # (another Trainer on the same solver may have registered its own)
            self.engine.set_train_forward_event(self._fwd_event, self._sumsq_host)

    def loss_and_grad(self, wavefields, h_states, k_sqs, residual, sources, num_iterations: Optional[int] = None, input_grads: bool = False):
        """The differentiable core of training_step (hybridnet.py:399-409) + backward: returns the engine's dict with the loss
        (1e4 * mean(res^2) over all unrolled residuals), the flat gradient (already reduced over ranks) and the n_steps lists."""
        T = int(self.solver.hparams.unrolling_steps if num_iterations is None else num_iterations)
        self.solver.set_source_maps(sources)   # :400
        self._reserve_sumsq(T * wavefields.shape[0])
        out = self.engine.train_grad(self.weights, wavefields.float().contiguous(), residual.float().contiguous(), h_states.float().contiguous(),
                                     k_sqs.float().contiguous(), sources.float().contiguous(), T, 1e4, input_grads, grad=self.grad)
        self.grad_reduce(self.grad)
        return out

    def optimizer_step(self):
        """on_after_backward (:172-176) + Adam (:250-258), on the device blob."""
        hp = self.solver.hparams
        self.step_count += 1
        self.engine.adam_step(self.weights, self.grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, (0.9, 0.95), 1e-8,
                              float(hp.weight_decay), float(hp.gradient_clip_val), self.trainable)

    def training_step(self, sos_batch: torch.Tensor, batch_idx: int = 0) -> dict:
        """hybridnet.py:385-505 without the TensorBoard logging: sample the buffer, unroll, loss, backward, optimiser step,
        then refill the sampled slots (advanced experience while it stays bounded and young, else a fresh map of ``sos_batch``)."""
        s, hp = self.solver, self.solver.hparams
        maxiter = min([self.current_epoch * 20 + 1, hp.max_iterations])
        wavefields, h_states, k_sqs, residual, sources, timesteps, indices = self.replaybuffer.sample(hp.batch_size)
        fwd_events = self.engine.counter("train_fwd_events")
        out = self.loss_and_grad(wavefields, h_states, k_sqs, residual, sources)
        fwd_recorded = self.engine.counter("train_fwd_events") == fwd_events + 1 and getattr(self.engine, "_fwd_sumsq", None) is self._sumsq_host
        loss = out["loss"][0]
        T = out["residuals"].shape[0]
        iteration = np.random.choice(T)
        res_it, wf_it, st_it = out["residuals"][iteration], out["wavefields"][iteration], out["states"][iteration]
        self.optimizer_step()
        # which sampled slots keep their (advanced) experience: bounded residual and young enough (:436-452).  The count of rejected slots decides how
        # many fresh maps Python's ``choice`` draws, as in the reference's loop, so the host needs the answer -- it waits for the FORWARD sweep only
        nb = wavefields.shape[0]
        if fwd_recorded:
            self._fwd_event.synchronize()
            self.engine.check_async_errors()   # (a side-stream hand-over of the forward sweep that timed out: the tape behind it is incomplete)
            meansq = self._sumsq_host[: T * nb].view(T, nb)[iteration].numpy() / np.float32(res_it[0].numel())   # res.pow(2).mean() per sample, fp32
        else:   # the library did not record the event for this call (a captured step, or another engine user replaced the registration): the reference's own reduction
            meansq = res_it.pow(2).mean(dim=(1, 2, 3)).cpu().numpy()
        new_timesteps = np.asarray(timesteps, dtype=np.int64) + iteration + 1
        keep = (meansq < 1) & (new_timesteps < maxiter)
        fresh = np.nonzero(~keep)[0]
        with torch.no_grad():
            # every sampled slot takes its advanced wavefield / hidden state / residual (k_sq and the source of a slot do not change: :441-449) with the index
            # tensor ``sample`` left on the device -- three launches, no gather, no copy; the rejected slots are then overwritten by fresh experiences
            self.replaybuffer.update(indices, np.where(keep, new_timesteps, 0), index_device=self.replaybuffer.last_sample_index,
                                     wavefield=wf_it, hidden_state=st_it, residual=res_it)
            if fresh.size:   # one random map of this batch per rejected slot (drawn in slot order, as the reference's loop does), solved from scratch
                maps = torch.stack([choice(sos_batch) for _ in fresh])
                k_sq, res, src = self._fresh_fields(maps, zeros=False)
                self.replaybuffer.update(indices[fresh], np.zeros(fresh.size, dtype=np.int64), zero=("wavefield", "hidden_state"), k_sq=k_sq, residual=res, source=src)
        counter = int(fresh.size)
        self.new_sos = counter
        self.global_step += 1
        self.epoch_losses.append(loss.detach())
        return {"loss": loss, "maxiter": maxiter, "unrolling": T, "new_sos": counter}

    def training_epoch_end(self) -> float:
        """hybridnet.py:379-383 + the scheduler step Lightning performs on ``train_loss_mean`` once per epoch (:276-281)."""
        mean = float(torch.stack(self.epoch_losses).mean()) if self.epoch_losses else float("nan")
        mean = allreduce_mean_scalar(mean, self.weights.device)   # every replica's scheduler monitors the same number
        self.epoch_losses = []
        self.scheduler.step(mean)
        self.lr = float(self._lr_holder.param_groups[0]["lr"])
        self.current_epoch += 1
        self.sync_to_module()
        return mean

    def fit(self, sos_train, max_epochs: int, steps_per_epoch: Optional[int] = None) -> List[float]:
        """A plain training loop over ``sos_train`` ([M, 1, N, N] tensor or dataset of [1, N, N] maps), batches of hparams.batch_size,
        ``drop_last=True`` as train_dataloader (:221-226).  Returns the epoch-mean losses."""
        hp = self.solver.hparams
        if not self.replaybuffer.filled.all():
            self.fill_replay_buffer(sos_train)
        n = len(sos_train)
        history = []
        for _ in range(max_epochs):
            order = np.arange(n)
            steps = n // hp.batch_size if steps_per_epoch is None else steps_per_epoch
            for i in range(steps):
                idx = order[(i * hp.batch_size) % max(n - hp.batch_size + 1, 1):][: hp.batch_size]
                batch = torch.stack([torch.as_tensor(sos_train[int(j)]) for j in idx]).to(self.solver.device)
                self.training_step(batch, i)
            history.append(self.training_epoch_end())
        return history
