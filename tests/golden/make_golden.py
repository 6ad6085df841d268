#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE.

Runs only in the build container, where the reference checkout is mounted
read-only at /root/reference.  The reference is imported UNMODIFIED; three of
its third-party imports are absent from the image (pytorch_lightning,
torchmetrics, cv2), so minimal stand-in modules are registered in
``sys.modules`` first.  Nothing of the reference is
copied: the outputs are data only (weights exported from the shipped
MIT-licensed checkpoint, setup tensors, input/output pairs, traces).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Inputs of the teacher-forced N=256 cases are NOT stored: they are re-created
from ``numpy.random.default_rng(seed)`` by ``tests/golden_inputs.py`` (the same
helper this script uses), so only the outputs / probes are committed.
"""
import collections
import inspect
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
CKPT = os.path.join(REF, "trained_models/jcp_paper_trained_weights.ckpt")
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)


def install_shims():
    class AttrDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    class LightningModule(nn.Module):
        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            args = inspect.getargvalues(frame)
            self.hparams = AttrDict({a: args.locals[a] for a in args.args if a != "self"})

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def freeze(self):
            for p in self.parameters():
                p.requires_grad = False
            self.eval()

        def log(self, *a, **k):
            pass

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, strict=True, **kwargs):
            torch.serialization.add_safe_globals(
                [(collections.OrderedDict, "pytorch_lightning.utilities.parsing.AttributeDict")]
            )
            ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            hp = dict(ck["hyper_parameters"])
            hp.update(kwargs)
            model = cls(**hp)
            model.load_state_dict(ck["state_dict"], strict=strict)
            return model

    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    tm = types.ModuleType("torchmetrics")
    tmr = types.ModuleType("torchmetrics.regression")

    class MeanAbsoluteError(nn.Module):
        def forward(self, a, b):
            return (a - b).abs().mean()

    tmr.MeanAbsoluteError = MeanAbsoluteError
    tm.regression = tmr
    sys.modules["torchmetrics"] = tm
    sys.modules["torchmetrics.regression"] = tmr
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, REF)


def load_reference():
    install_shims()
    from helmnet import IterativeSolver  # noqa: the reference, unmodified

    solver = IterativeSolver.load_from_checkpoint(CKPT, strict=False, test_data_path=None)
    solver.freeze()
    return solver


def npf(t):
    return t.detach().cpu().numpy().astype(np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from golden_inputs import (teacher_inputs, readme_sos, ring_sos_batch)  # tests/golden_inputs.py

    solver = load_reference()
    # ---- 1. weights + hparams ------------------------------------------------
    sd = solver.state_dict()
    weights = {k[2:]: npf(v) for k, v in sd.items() if k.startswith("f.")}
    assert sum(v.size for v in weights.values()) == 48160
    np.savez(os.path.join(HERE, "jcp_weights.npz"), **weights)
    hp = {k: v for k, v in dict(solver.hparams).items()}
    with open(os.path.join(HERE, "jcp_hparams.json"), "w") as f:
        json.dump(hp, f, indent=1, sort_keys=True)

    with torch.no_grad():
        # ---- 2. setup tensors -----------------------------------------------
        setup = {}
        for n, loc in ((96, [82, 48]), (256, [30, 128]), (512, [450, 256])):
            solver.set_domain_size(n, source_location=loc)
            L = solver.Lap
            setup[f"n{n}_kx_row"] = npf(L.kx[0, 0, :, 1])       # i*kx along W
            setup[f"n{n}_ky_col"] = npf(L.ky[0, :, 0, 1])
            setup[f"n{n}_kxsq_row"] = npf(L.kx_sq[0, 0, :, 0])
            setup[f"n{n}_kysq_col"] = npf(L.ky_sq[0, :, 0, 0])
            for nm in ("ax", "bx"):
                setup[f"n{n}_{nm}_row"] = npf(getattr(L, nm)[0, 0, :, :])   # [N,2] varies along W
                # the x tables must not vary along H
                assert torch.equal(getattr(L, nm)[0, 0], getattr(L, nm)[0, n // 3])
            for nm in ("ay", "by"):
                setup[f"n{n}_{nm}_col"] = npf(getattr(L, nm)[0, :, 0, :])   # [N,2] varies along H
                assert torch.equal(getattr(L, nm)[0, :, 0], getattr(L, nm)[0, :, n // 3])
            setup[f"n{n}_sigma_x_row"] = npf(solver.sigmas[0, 0, :])
            setup[f"n{n}_sigma_y_col"] = npf(solver.sigmas[1, :, 0])
            src = solver.source
            setup[f"n{n}_source_peak"] = npf(src[0, :, loc[0], loc[1]])
            s2 = src.clone()
            s2[0, :, loc[0], loc[1]] = 0
            setup[f"n{n}_source_offpeak_absmax"] = np.float32(s2.abs().max().item())
            if n == 96:
                setup["n96_source"] = npf(src)
                setup["n96_ax_full"] = npf(L.ax)
                setup["n96_by_full"] = npf(L.by)
        np.savez_compressed(os.path.join(HERE, "setup.npz"), **setup)

        # ---- 3. teacher-forced operator pairs --------------------------------
        tf = {}
        for n, loc, b in ((96, [82, 48], 2), (256, [30, 128], 2), (512, [450, 256], 1)):
            solver.set_domain_size(n, source_location=loc)
            ti = teacher_inputs(n, b, seed=1000 + n)
            wf, res, st, sos = (torch.from_numpy(ti[k]) for k in ("wf", "res", "states", "sos"))
            k_sq, _ = solver.get_initials(sos)
            lap = solver.apply_laplacian(wf)
            r = solver.get_residual(wf, k_sq)
            solver.f.set_states(st, flatten=True)
            sig = solver.sigmas.unsqueeze(0).repeat(b, 1, 1, 1)
            d = solver.f(torch.cat([wf, 1e3 * res, sig], 1))
            st_new = solver.f.get_states(flatten=True)
            solver.f.set_states(st, flatten=True)
            wf2, res2 = solver.single_step(wf, k_sq, res)
            out = dict(lap=lap, residual=r, unet_d=d, states_new=st_new, step_wf=wf2, step_res=res2)
            for k, v in out.items():
                v = npf(v.contiguous())
                tf[f"n{n}_{k}_absmax"] = np.float32(np.abs(v).max())
                tf[f"n{n}_{k}_sum"] = np.float64(v.astype(np.float64).sum())
                if n == 96:
                    tf[f"n{n}_{k}"] = v
                elif v.ndim == 4:
                    tf[f"n{n}_{k}_crop"] = v[:, :, :40, :40].copy()       # PML corner + interior
                    tf[f"n{n}_{k}_stride"] = v[:, :, 3::7, 5::11].copy()  # whole-domain probe
                else:
                    tf[f"n{n}_{k}_stride"] = v[:, :, 1::37].copy()
        np.savez_compressed(os.path.join(HERE, "teacher_forced.npz"), **tf)

        # ---- 4. free-running solves ------------------------------------------
        fr = {}

        def run(tag, sos, n, loc=None, src_map=None, iters=100, keep_at=()):
            if loc is not None:
                solver.set_domain_size(n, source_location=loc)
            else:
                solver.set_domain_size(n, source_map=src_map)
            k_sq, wf = solver.get_initials(sos)
            solver.f.clear_states(wf)
            res = solver.get_residual(wf, k_sq)
            trace = []
            for it in range(iters):
                wf, res = solver.single_step(wf, k_sq, res)
                trace.append(npf(solver.test_loss_function(res)))
                if it + 1 in keep_at:
                    fr[f"{tag}_wf_it{it + 1}"] = npf(wf)
            fr[f"{tag}_rmse"] = np.stack(trace)
            fr[f"{tag}_wf_absmax"] = np.float32(wf.abs().max().item())

        # config 1: homogeneous 256^2, source [30,128], 100 iterations
        run("cfg1", torch.ones(1, 1, 256, 256), 256, loc=[30, 128], iters=100, keep_at=(1, 100))
        # README problem (test.py:14-21), 300 iterations
        run("readme", torch.from_numpy(readme_sos()), 256, loc=[30, 128], iters=300, keep_at=(100, 300))
        # examples/simple_scattering.py:16-36 problem (line source map), 100 iterations
        sos = np.ones((256, 256), np.float32)
        sos[100:170, 30:240] = 1.5
        smap = np.zeros((1, 2, 256, 256), np.float32)
        smap[0, 0, 30, 120:130] = 1
        run("scatter", torch.from_numpy(sos)[None, None], 256, src_map=torch.from_numpy(smap), iters=100, keep_at=(100,))
        # native training size, ring phantoms, batch 3, 200 iterations
        run("ring96", torch.from_numpy(ring_sos_batch(96, 3, seed=7)), 96, loc=[82, 48], iters=200, keep_at=(50, 200))
        np.savez_compressed(os.path.join(HERE, "free_run.npz"), **fr)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
