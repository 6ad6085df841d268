"""SoS-map datasets (host side): mirror of the reference's ``EllipsesDataset`` / ``get_dataset``.

Reference helmnet/dataloaders.py:9-24 (get_dataset), :27-80 (dataset container), :82-156 (the random
harmonic "skull" ring).  The ring is rasterised by helmnet_amd.phantoms.ring_sos (no cv2).  Only what
feeds the inference path is mirrored: make / load / save / tensor conversion / indexing.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from .phantoms import ring_sos


class EllipsesDataset(Dataset):
    """Dataset of oversimplified skulls: [num, size, size] float32 maps, background 1.0, ring 1.5..2.0."""

    def __init__(self):
        self._all_sos = []
        self.all_sos_numpy = []

    @property
    def all_sos(self):
        if len(self._all_sos) == 0:
            return []
        return self._all_sos

    def make_dataset(self, num_ellipses: int = 5000, imsize: int = 128, seed: int = 0):
        rng = np.random.default_rng(seed)
        self.all_sos_numpy = np.stack([ring_sos(imsize, rng) for _ in range(num_ellipses)], axis=0)

    def load_dataset(self, filepath: str = "data/ellipses.npy"):
        self.all_sos_numpy = np.array(np.load(filepath), np.float32)

    def save_dataset(self, filepath: str):
        np.save(filepath, self.all_sos_numpy)

    def sos_maps_to_tensor(self):
        """[num, 1, size, size] float32 tensor, the layout IterativeSolver.forward expects."""
        self._all_sos = torch.from_numpy(np.asarray(self.all_sos_numpy)).unsqueeze(1).float()

    def __len__(self):
        return len(self._all_sos)

    def __getitem__(self, idx):
        return self._all_sos[idx]


def get_dataset(dataset_path: str, source_location: str = "cuda:7", destination: str = "cpu") -> Dataset:
    """Load a pickled dataset (`.ph`, torch.save of an EllipsesDataset) or a `.npy` array of maps."""
    if dataset_path.endswith(".npy"):
        ds = EllipsesDataset()
        ds.load_dataset(dataset_path)
        ds.sos_maps_to_tensor()
        return ds
    return torch.load(dataset_path, map_location={source_location: destination}, weights_only=False)
