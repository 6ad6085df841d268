"""Training half of the reference's ``IterativeSolver`` on the HIP library.

Mirrors, without Lightning: ``helmnet/replaybuffer.py:1-47`` (``Experience`` / ``ReplayBuffer``), the buffer fill of
``train_dataloader`` (hybridnet.py:192-218), ``training_step`` (:385-505), ``on_after_backward`` (:172-176),
``configure_optimizers`` (:250-283: Adam(0.9, 0.95) + ReduceLROnPlateau on the epoch-mean training loss) and
``training_epoch_end`` (:379-383).  The arithmetic -- ten unrolled solver iterations, the loss, back-propagation through
time, gradient clipping and the Adam update -- runs in libhelmnet_hip.so (``hn_train_grad`` / ``hn_adam_step``); this
module holds the flat parameter / gradient / moment tensors, the replay buffer and the host-side bookkeeping.

Data-parallel training (the reference trains with Lightning DDP, train.py:103-112): one process per GPU, every rank
draws its own replay-buffer batch, the flat gradient (one 193 KB bucket) is averaged with a single all-reduce over
RCCL between ``hn_train_grad`` and ``hn_adam_step`` (``allreduce_gradients``), so the replicas stay bit-identical.
"""
from __future__ import annotations

import collections
from random import choice
from typing import List, Optional

import numpy as np
import torch

from .engine import pack_weights, unpack_weights, weight_shapes

# replaybuffer.py:8-18
Experience = collections.namedtuple(
    "Experience", field_names=["wavefield", "hidden_state", "k_sq", "residual", "source", "iteration"]
)


class ReplayBuffer:
    """replaybuffer.py:20-47: a fixed-capacity list of Experiences addressed by index; ``sample`` draws ``batch_size``
    distinct slots with ``np.random.choice`` and stacks the fields."""

    def __init__(self, capacity: int):
        self.buffer = [None for _ in range(capacity)]
        self.capacity = capacity

    def __len__(self):
        return self.capacity

    def append(self, experience, index):
        self.buffer[index] = experience

    def sample(self, batch_size: int):
        if batch_size > self.capacity:
            batch_size = self.capacity
        indices = np.random.choice(self.capacity, batch_size, replace=False)
        wavefields, h_states, k_sqs, residual, source, iterations = zip(*[self.buffer[t] for t in indices])
        return (torch.stack(wavefields, 0), torch.stack(h_states, 0), torch.stack(k_sqs, 0), torch.stack(residual, 0),
                torch.stack(source, 0), iterations, indices)


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
        self.step_count = 0
        self.current_epoch = 0
        self.global_step = 0
        self.lr = float(hp.learning_rate)
        # ReduceLROnPlateau(mode="min", factor=0.5, patience=10, min_lr) monitoring the epoch-mean training loss (:270-283);
        # torch's scheduler does the bookkeeping on a one-parameter stand-in optimiser whose lr is read back
        self._lr_holder = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=self.lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self._lr_holder, mode="min", factor=0.5, patience=10,
                                                                     min_lr=float(hp.minimum_learning_rate))
        self.replaybuffer = ReplayBuffer(hp.buffer_size)
        self.epoch_losses: List[torch.Tensor] = []
        self.new_sos = 0

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
        self.scheduler.load_state_dict(sd["scheduler"])
        self._lr_holder.param_groups[0]["lr"] = self.lr
        self.sync_to_module()

    # ------------------------------------------------------------------ replay buffer -----------
    def _fresh_experience(self, sos_map: torch.Tensor, iteration: int) -> Experience:
        """Initial (zero wavefield) experience of one [1, 1, N, N] sound-speed map (hybridnet.py:203-216, 454-462)."""
        s = self.solver
        s.reset_source()
        k_sq, wf = s.get_initials(sos_map.float().to(s.device))
        s.f.clear_states(wf)
        h = s.f.get_states(flatten=True)
        # after set_source_maps(sources) the reference's self.source holds one map per sample and its get_residual broadcasts
        # (:556); the maps of a batch are equal unless the caller set different ones -- the first one is the source of a fresh slot
        src0 = s.source[:1].detach().float().contiguous()
        res = s.engine().residual(wf, k_sq.contiguous(), src0)
        return Experience(wf[0], h[0], k_sq[0], res[0], src0[0], iteration)

    def fill_replay_buffer(self, sos_train):
        """hybridnet.py:192-218: one fresh experience per slot, slot ``c`` starting at nominal iteration ``10 c``."""
        with torch.no_grad():
            for counter in range(len(self.replaybuffer)):
                sos_map = torch.as_tensor(sos_train[counter]).unsqueeze(0)
                self.replaybuffer.append(self._fresh_experience(sos_map, counter * 10), counter)

    # ------------------------------------------------------------------ one step ----------------
    def loss_and_grad(self, wavefields, h_states, k_sqs, residual, sources, num_iterations: Optional[int] = None, input_grads: bool = False):
        """The differentiable core of training_step (hybridnet.py:399-409) + backward: returns the engine's dict with the loss
        (1e4 * mean(res^2) over all unrolled residuals), the flat gradient (already reduced over ranks) and the n_steps lists."""
        T = int(self.solver.hparams.unrolling_steps if num_iterations is None else num_iterations)
        self.solver.set_source_maps(sources)   # :400
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
        out = self.loss_and_grad(wavefields, h_states, k_sqs, residual, sources)
        self.optimizer_step()
        loss = out["loss"][0]
        T = out["residuals"].shape[0]
        iteration = np.random.choice(T)
        res_it, wf_it, st_it = out["residuals"][iteration], out["wavefields"][iteration], out["states"][iteration]
        keep = (res_it.pow(2).mean((1, 2, 3)) < 1).cpu().numpy()      # one device-to-host read for the whole batch
        counter = 0
        for sample_idx in range(wavefields.shape[0]):
            new_timesteps = timesteps[sample_idx] + iteration + 1
            if keep[sample_idx] and new_timesteps < maxiter:
                exp = Experience(wf_it[sample_idx].clone(), st_it[sample_idx].clone(), k_sqs[sample_idx], res_it[sample_idx].clone(),
                                 sources[sample_idx], new_timesteps)
            else:
                with torch.no_grad():
                    exp = self._fresh_experience(choice(sos_batch).unsqueeze(0), 0)
                counter += 1
            self.replaybuffer.append(exp, indices[sample_idx])
        self.new_sos = counter
        self.global_step += 1
        self.epoch_losses.append(loss.detach())
        return {"loss": loss, "maxiter": maxiter, "unrolling": T, "new_sos": counter}

    def training_epoch_end(self) -> float:
        """hybridnet.py:379-383 + the scheduler step Lightning performs on ``train_loss_mean`` once per epoch (:276-281)."""
        mean = float(torch.stack(self.epoch_losses).mean()) if self.epoch_losses else float("nan")
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
        if self.replaybuffer.buffer[0] is None:
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
