"""Read the reference's PyTorch-Lightning checkpoints without PyTorch-Lightning.

The shipped ``trained_models/jcp_paper_trained_weights.ckpt`` is a Lightning-0.9 pickle whose
``hyper_parameters`` entry is a ``pytorch_lightning.utilities.parsing.AttributeDict``
(SURVEY.md section 5).  ``torch.load(weights_only=True)`` accepts it once that class name is
mapped onto ``collections.OrderedDict``.
"""
from __future__ import annotations

import collections
import json
import os
from typing import Dict, Tuple

import numpy as np
import torch


class AttributeDict(dict):
    """dict with attribute access (what callers expect of ``solver.hparams``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def read_lightning_checkpoint(path: str, map_location="cpu") -> Tuple[Dict, Dict[str, torch.Tensor]]:
    """-> (hyper_parameters dict, state_dict)."""
    torch.serialization.add_safe_globals(
        [(collections.OrderedDict, "pytorch_lightning.utilities.parsing.AttributeDict")]
    )
    ck = torch.load(path, map_location=map_location, weights_only=True)
    if "state_dict" not in ck:
        raise ValueError(f"{path} is not a Lightning checkpoint (no 'state_dict')")
    hp = ck.get("hyper_parameters", ck.get("hparams", {}))
    return dict(hp), ck["state_dict"]


def read_exported_weights(npz_path: str, hparams_json: str) -> Tuple[Dict, Dict[str, torch.Tensor]]:
    """The `f.*` tensors + hparams exported from the shipped (MIT-licensed) checkpoint, carried as package
    data (helmnet_amd/data/jcp_*); used where the reference checkout (and so the .ckpt) is not available,
    e.g. the GPU box."""
    with open(hparams_json) as f:
        hp = json.load(f)
    with np.load(npz_path) as z:
        sd = {"f." + k: torch.from_numpy(z[k]) for k in z.files}
    return hp, sd


def default_exported_weights() -> Tuple[str, str]:
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
    return os.path.join(root, "jcp_weights.npz"), os.path.join(root, "jcp_hparams.json")
