"""Drop-in replacement for the reference's ``IterativeSolver`` inference API on MI355X.

Mirrors reference helmnet/hybridnet.py (inference half): ``load_from_checkpoint`` (Lightning
classmethod, re-implemented without Lightning), ``hparams``, ``freeze``, ``device``, ``to``,
``set_domain_size`` (:92-108), ``set_laplacian`` (:110-131), ``setup_source / set_source /
set_source_maps / set_multiple_sources / reset_source`` (:133-170), ``get_initials``
(:522-538), ``apply_laplacian`` (:540-542), ``get_residual`` (:544-556), ``single_step``
(:558-584), ``n_steps`` (:586-623), ``forward`` (:654-697), ``forward_variable_src``
(:699-754), ``test_loss_function`` (:295-297), ``loss_function`` (:285-293).  The training half (replay buffer,
``training_step``, optimiser, scheduler) lives in ``helmnet_amd.training`` (``solver.trainer()``), on ``hn_train_grad`` /
``hn_adam_step``; Lightning's hooks and TensorBoard logging have no counterpart.

All per-iteration arithmetic runs in libhelmnet_hip.so: ``forward`` / ``n_steps`` hand the whole
loop to ``hn_step`` (fused HIP kernels), ``get_residual`` to ``hn_residual``, ``f`` to ``hn_unet``.
There is no CPU path: a solver left on the CPU raises as soon as it is asked to compute.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .checkpoint import AttributeDict, read_lightning_checkpoint
from .engine import Engine
from .laplacian import FastLaplacianWithPML
from .source import SourceModule
from .unet import HybridNet


class IterativeSolver(nn.Module):
    def __init__(
        self,
        domain_size: int,
        k: float,
        omega: float,
        PMLsize: int,
        sigma_max: float,
        source_location: list,
        train_data_path: Optional[str] = None,
        validation_data_path: Optional[str] = None,
        test_data_path: Optional[str] = None,
        activation_function: str = "relu",
        architecture: str = "custom_unet",
        gradient_clip_val: int = 0,
        batch_size: int = 24,
        buffer_size: int = 1000,
        depth: int = 4,
        features: int = 8,
        learning_rate: float = 1e-4,
        loss: str = "mse",
        minimum_learning_rate: float = 1e-4,
        optimizer: str = "adam",
        weight_decay: float = 0.0,
        max_iterations: int = 100,
        source_amplitude: int = 10,
        source_phase: int = 0,
        source_smoothing: bool = False,
        state_channels: int = 2,
        state_depth: int = 4,
        unrolling_steps: int = 10,
    ):
        super().__init__()
        hp = dict(locals())
        hp.pop("self")
        hp.pop("__class__", None)
        self.hparams = AttributeDict(hp)  # save_hyperparameters() equivalent (hybridnet.py:54)
        self._engine: Optional[Engine] = None
        self._unet_precision = None   # None: the library default (fp32, or HN_UNET_IMPL at context creation)
        self.register_buffer("sigmas", None)
        self.set_laplacian()
        self.setup_source()
        self.init_f()

        def weights_init(m):  # hybridnet.py:70-75 (overwritten by load_state_dict)
            if isinstance(m, nn.Conv2d):
                torch.nn.init.xavier_normal_(m.weight, gain=0.02)

        self.f.apply(weights_init)

    # ------------------------------------------------------------------ construction -----
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = True, **kwargs):
        """Lightning's classmethod: hyper_parameters (+ overrides) -> cls(**hp) -> load_state_dict."""
        hp, sd = read_lightning_checkpoint(checkpoint_path, map_location or "cpu")
        hp.update(kwargs)
        model = cls(**hp)
        model.load_state_dict(sd, strict=strict)
        return model

    @classmethod
    def from_exported_weights(cls, npz_path: Optional[str] = None, hparams_json: Optional[str] = None, **kwargs):
        """Build from the `f.*` tensors exported out of the shipped checkpoint (package data, helmnet_amd/data)."""
        from .checkpoint import default_exported_weights, read_exported_weights
        d_npz, d_json = default_exported_weights()
        hp, sd = read_exported_weights(npz_path or d_npz, hparams_json or d_json)
        hp.update(kwargs)
        model = cls(**hp)
        model.load_state_dict(sd, strict=False)
        return model

    def init_f(self):
        if self.hparams.architecture != "custom_unet":
            raise NotImplementedError("Unknown architecture {}".format(self.hparams.architecture))
        self.f = HybridNet(
            activation_function=self.hparams.activation_function,
            depth=self.hparams.depth,
            domain_size=self.hparams.domain_size,
            features=self.hparams.features,
            inchannels=6,
            state_channels=self.hparams.state_channels,
            state_depth=self.hparams.state_depth,
        )

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    @property
    def device(self) -> torch.device:
        return self.source.device

    # ------------------------------------------------------------------ engine -----------
    def engine(self) -> Engine:
        """The library context for this solver's device, with domain + weights in sync."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError(
                "IterativeSolver is on the CPU: helmnet_amd computes on MI355X only -- call "
                "solver.to('cuda:0') first (the CPU restatement lives in oracle/ for tests)."
            )
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if self._engine is None or self._engine.device != dev:
            self._engine = Engine(dev)
        key = (int(self.hparams.domain_size), int(self.hparams.PMLsize), float(self.hparams.sigma_max), float(self.hparams.k))
        if self._engine.domain_key != key:
            self._engine.set_domain(*key)
        if self._unet_precision is not None and self._engine.unet_precision != self._unet_precision:
            self._engine.set_unet_precision(self._unet_precision)
        self.f.bind(self._engine)
        self.Lap.bind(self._engine)
        self.f.sync_weights(self._engine)
        return self._engine

    def set_unet_precision(self, mode: str):
        """Extension (BASELINE.json configs[4]): arithmetic of the UNet convolutions -- 'fp32' (the reference's,
        default), 'fp16' (mixed fp16 UNet / fp32 spectral residual), 'bf16x3' / 'bf16x2' (split-bf16 emulations),
        'valu'.  Per solver (hn_set_unet_precision on its context), not per process."""
        from . import _lib
        if mode not in _lib.HN_PRECISION:
            raise ValueError(f"unknown UNet precision {mode!r} (choose from {sorted(_lib.HN_PRECISION)})")
        self._unet_precision = mode
        if self._engine is not None:
            self._engine.set_unet_precision(mode)

    # ------------------------------------------------------------------ setup ------------
    def set_domain_size(self, domain_size, source_location=None, source_map=None):
        self.hparams.domain_size = domain_size
        self.f.domain_size = self.hparams.domain_size
        self.set_laplacian()
        # the reference builds the source module at hparams.source_location first and only then
        # moves it (hybridnet.py:96,101-104), which indexes out of bounds when the checkpoint's
        # location does not fit the new domain; build it at the requested location directly.
        self.setup_source(location=source_location)
        self.Lap.to(self.device)
        self.source_module.to(self.device)
        if source_location is not None:
            self.set_multiple_sources([source_location])
        else:
            self.set_source_maps(source_map)
        self.f.init_by_size()
        for enc, size in zip(self.f.enc, self.f.states_dimension):
            enc.domain_size = size
            enc.state = None

    def set_laplacian(self):
        dev = self.source.device if hasattr(self, "source") else torch.device("cpu")
        self.Lap = FastLaplacianWithPML(
            domain_size=self.hparams.domain_size,
            PMLsize=self.hparams.PMLsize,
            k=self.hparams.k,
            sigma_max=self.hparams.sigma_max,
        )
        sx, sy = self.Lap.sigmas()
        self.sigmas = torch.stack([sx.detach().clone(), sy.detach().clone()]).float().to(dev)

    def setup_source(self, location=None):
        n = self.hparams.domain_size
        if location is None:
            location = self.hparams.source_location
            if not (0 <= location[0] < n and 0 <= location[1] < n):
                location = [n // 2, n // 2]  # placeholder; a source map / location is set right after
        self.source_module = SourceModule(
            image_size=n,
            omega=self.hparams.omega,
            location=location,
            amplitude=self.hparams.source_amplitude,
            phase=self.hparams.source_phase,
            smooth=self.hparams.source_smoothing,
        )
        with torch.no_grad():
            self.set_source()

    def set_source_maps(self, sourceval):
        dev = self.source.device if hasattr(self, "source") else sourceval.device
        self.source = nn.Parameter(sourceval.to(dev), requires_grad=False)

    def set_source(self):
        self.set_source_maps(self.source_module.spatial_map(0).permute(0, 3, 1, 2).contiguous())

    def reset_source(self):
        with torch.no_grad():
            if not self.source_module.get_location() == self.hparams.source_location:
                self.source_module.set_new_location(self.hparams.source_location)
                self.set_source()

    def set_multiple_sources(self, source_locations):
        maps = []
        with torch.no_grad():
            for loc in source_locations:
                self.source_module.set_new_location(loc)
                maps.append(self.source_module.spatial_map(0).permute(0, 3, 1, 2))
            self.set_source_maps(torch.cat(maps, 0).contiguous())

    # ------------------------------------------------------------------ operators --------
    @staticmethod
    def test_loss_function(x):
        """Per-sample residual RMSE (hybridnet.py:295-297); plain tensor reduction for callers
        that hold a residual tensor -- the solver loop itself uses the fused hn_step norms."""
        return x.pow(2).mean((1, 2, 3)).sqrt()

    def loss_function(self, x):
        """hybridnet.py:285-293: mean square over every element ('mse' is the only loss the reference implements)."""
        if self.hparams.loss == "mse":
            return x.pow(2).mean()
        raise NotImplementedError("The loss function {} is not implemented".format(self.hparams.loss))

    def get_random_source_loc(self):
        """hybridnet.py:178-190: a random source location on a circle of radius L - PMLsize - 2 around the domain centre."""
        import numpy as np
        theta = 2 * np.pi * np.random.rand()
        L = self.hparams.domain_size // 2
        dL = L - self.hparams.PMLsize - 2
        return [int(L + dL * np.cos(theta)), int(L + dL * np.sin(theta))]

    def validation_step(self, batch, batch_idx=0):
        """hybridnet.py:333-352: one random source per sample, max_iterations solver iterations, loss = sqrt(mean(res_last^2)) over the
        batch (NaN -> inf); returns the loss, the first sample's wavefield mapped to [0, 1] and the batch index."""
        self.set_multiple_sources([self.get_random_source_loc() for _ in range(batch.shape[0])])
        output = self.forward(batch, num_iterations=self.hparams.max_iterations, return_wavefields=False, return_states=False, residuals="last")
        loss = self.loss_function(output["residuals"][-1]).sqrt()
        loss = torch.where(torch.isnan(loss), torch.full_like(loss, float("inf")), loss)
        sample_wavefield = (torch.nn.functional.hardtanh(output["wavefields"][0][0]) + 1) / 2
        return {"loss": loss, "sample_wavefield": sample_wavefield, "batch_idx": batch_idx}

    def test_step(self, batch, batch_idx=0):
        """hybridnet.py:299-314: max_iterations iterations with every wavefield kept; per-sample residual RMSE after every iteration
        ([B, max_iterations], from the fused on-device norms) and the list of wavefields."""
        self.reset_source()
        output = self.forward(batch, num_iterations=self.hparams.max_iterations, return_wavefields=True, return_states=False, residuals="norms")
        return {"losses": output["residual_norms"].transpose(0, 1).contiguous(), "wavefields": list(output["wavefields"])}

    def trainer(self, **kwargs):
        """The training half of the reference class (replay buffer, training_step, Adam + ReduceLROnPlateau) for this solver."""
        from .training import Trainer
        return Trainer(self, **kwargs)

    def get_initials(self, sos_maps: torch.Tensor):
        k_sq = (self.hparams.omega / sos_maps) ** 2
        wavefield = torch.zeros(k_sq.shape[0], 2, k_sq.shape[2], k_sq.shape[3], device=k_sq.device)
        return k_sq, wavefield

    def _src(self) -> torch.Tensor:
        return self.source.detach().float().contiguous()

    def apply_laplacian(self, x: torch.Tensor):
        return self.engine().laplacian(x.float().contiguous())

    def get_residual(self, x: torch.Tensor, k_sq: torch.Tensor):
        return self.engine().residual(x.float().contiguous(), k_sq.float().contiguous(), self._src())

    def single_step(self, wavefield, k_sq, residual, get_residual: bool = True):
        """One iteration on caller-held tensors (hybridnet.py:558-584); network states live in
        ``self.f`` as in the reference.  Inputs are not modified."""
        eng = self.engine()
        if any(enc.state is None for enc in self.f.enc):
            raise ValueError("You must set or clear the state before using this module")
        wf = wavefield.detach().float().clone().contiguous()
        res = residual.detach().float().clone().contiguous()
        st = self.f.get_states(flatten=True).float().contiguous().clone()
        keep = self.f.to_engine_states(st)
        eng.step(wf, res, st, k_sq.float().contiguous(), self._src(), 1)
        self.f.from_engine_states(keep, st)
        self.f.adopt_states(st)
        return (wf, res) if get_residual else wf

    # ------------------------------------------------------------------ loops ------------
    def _run(self, wf, res, st, k_sq, num_iterations, return_wavefields, return_states, residuals: str):
        eng = self.engine()
        b, n, K = wf.shape[0], wf.shape[-1], int(num_iterations)
        dev = wf.device

        def hist(shape, what):
            try:
                return torch.empty(shape, device=dev, dtype=torch.float32)
            except RuntimeError as e:  # out of memory
                raise RuntimeError(
                    f"cannot keep {what} for {K} iterations ({shape}); pass residuals='norms' or 'last' "
                    "to IterativeSolver.forward, or fewer iterations"
                ) from e

        res_hist = hist((K, b, 2, n, n), "every residual") if residuals == "all" and K > 0 else None
        wf_hist = hist((K, b, 2, n, n), "every wavefield") if return_wavefields and K > 0 else None
        st_hist = hist((K, b, 2, eng.state_len), "every hidden state") if return_states and K > 0 else None
        rmse = torch.empty((K, b), device=dev, dtype=torch.float32) if K > 0 else None
        if K > 0:
            keep = self.f.to_engine_states(st)      # levels without state: zeros in, the caller's values back out
            eng.step(wf, res, st, k_sq, self._src(), K, res_hist, wf_hist, st_hist, rmse)
            self.f.from_engine_states(keep, st, st_hist)
        self.f.adopt_states(st)
        out = {
            "wavefields": list(wf_hist.unbind(0)) if wf_hist is not None else [wf],
            "residuals": list(res_hist.unbind(0)) if res_hist is not None else ([res] if residuals == "last" else []),
            "states": list(st_hist.unbind(0)) if st_hist is not None else [],
            "last_iteration": K - 1,
            "residual_norms": rmse,  # [K, B] per-sample RMSE (extension; the reference derives it afterwards)
        }
        if residuals == "norms":
            out["residuals"] = []
            out["last_residual"] = res
        return out

    def forward(self, sos_maps, return_wavefields=False, return_states=False, num_iterations=None,
                stop_if_diverge=False, residuals: str = "all"):
        """hybridnet.py:654-697.  ``residuals``: "all" keeps every residual tensor like the
        reference (K x B x 2 x N x N floats), "norms" keeps only the per-iteration per-sample RMSE
        (``out["residual_norms"]``), "last" only the final residual."""
        if residuals not in ("all", "norms", "last"):
            raise ValueError("residuals must be 'all', 'norms' or 'last'")
        if num_iterations is None:
            num_iterations = self.hparams.max_iterations
        sos_maps = sos_maps.float().contiguous()
        k_sq, wf = self.get_initials(sos_maps)
        self.f.clear_states(wf)
        res = self.get_residual(wf, k_sq)
        st = self.f.get_states(flatten=True).contiguous()
        return self._run(wf, res, st, k_sq.contiguous(), num_iterations, return_wavefields, return_states, residuals)

    def n_steps(self, wavefield, k_sq, residual, num_iterations, return_wavefields=False, return_states=False,
                residuals: str = "all"):
        """hybridnet.py:586-623: continue from given wavefield / residual and the states held in f."""
        wf = wavefield.detach().float().clone().contiguous()
        res = residual.detach().float().clone().contiguous()
        st = self.f.get_states(flatten=True).float().contiguous().clone()
        return self._run(wf, res, st, k_sq.float().contiguous(), num_iterations, return_wavefields, return_states, residuals)

    def solve_to_tolerance(self, sos_maps, tol: float, max_iterations: int = None, check_every: int = 50,
                           norm_reduce=None) -> dict:
        """Extension (BASELINE.json configs[4], "convergence-to-tolerance"): iterate in chunks of
        ``check_every`` until the WORST per-sample residual RMSE (hybridnet.py:295-297) is below ``tol`` or
        ``max_iterations`` is reached.  One device-to-host read of a single float per chunk; with
        ``norm_reduce`` (e.g. helmnet_amd.distributed.allreduce_residual_norms) the test is global over ranks.
        Returns wavefield, last residual, per-iteration RMSE trace [K, B], iterations run and whether it converged."""
        if max_iterations is None:
            max_iterations = self.hparams.max_iterations
        sos_maps = sos_maps.float().contiguous()
        k_sq, wf = self.get_initials(sos_maps)
        self.f.clear_states(wf)
        res = self.get_residual(wf, k_sq)
        k_sq = k_sq.contiguous()
        done, traces, converged = 0, [], False
        while done < max_iterations:
            chunk = min(int(check_every), max_iterations - done)
            out = self.n_steps(wf, k_sq, res, chunk, residuals="norms")
            wf, res = out["wavefields"][0], out["last_residual"]
            traces.append(out["residual_norms"])
            done += chunk
            worst = out["residual_norms"][-1].max() if norm_reduce is None else norm_reduce(out["residual_norms"][-1], "max")
            if float(worst) < tol:
                converged = True
                break
        return {"wavefield": wf, "residual": res, "residual_norms": torch.cat(traces, 0), "iterations": done, "converged": converged}

    def forward_variable_src(self, sos_maps, src_time_pairs, return_wavefields=False, return_states=False,
                             num_iterations=None, stop_if_diverge=False, residuals: str = "all"):
        """hybridnet.py:699-754: swap the source map at given iterations (the residual is recomputed
        with the new source before the step, states and wavefield carry over)."""
        if num_iterations is None:
            num_iterations = self.hparams.max_iterations
        times = list(src_time_pairs["iteration"])
        maps = iter(src_time_pairs["src_maps"])
        sos_maps = sos_maps.float().contiguous()
        k_sq, wf = self.get_initials(sos_maps)
        self.f.clear_states(wf)
        res = self.get_residual(wf, k_sq)
        st = self.f.get_states(flatten=True).contiguous()
        cuts = sorted(set(t for t in times if 0 <= t < num_iterations) | {0, num_iterations})
        merged = {"wavefields": [], "residuals": [], "states": [], "last_iteration": num_iterations - 1}
        norms: List[torch.Tensor] = []
        for a, bnd in zip(cuts[:-1], cuts[1:]):
            if a in times:
                self.set_source_maps(next(maps))
                res = self.get_residual(wf, k_sq)
            part = self._run(wf, res, st, k_sq, bnd - a, return_wavefields, return_states, residuals)
            if residuals == "all":
                merged["residuals"] += part["residuals"]
            merged["states"] += part["states"]
            if return_wavefields:
                merged["wavefields"] += part["wavefields"]
            norms.append(part["residual_norms"])
        if not return_wavefields:
            merged["wavefields"].append(wf)
        if residuals == "last":      # wf / res / st are updated in place by every segment: only the final ones are kept
            merged["residuals"] = [res]
        elif residuals == "norms":
            merged["last_residual"] = res
        merged["residual_norms"] = torch.cat(norms, 0) if norms else None
        return merged
