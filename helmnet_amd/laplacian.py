"""Host-side mirror of the reference's spectral operator module.

Mirrors ``FastLaplacianWithPML`` (reference helmnet/spectral.py:246-363): same constructor,
same attribute names and tensor layouts (``kx, ky, kx_sq, ky_sq, ax, bx, ay, by`` as
``[1, N, N, 2]`` (re, im) fp32 parameters, ``sigma_x, sigma_y`` as ``[N, N]``), same
``forward(x[B, N, N, 2])`` and ``sigmas()``.  The tables are built exactly as the reference does
(float64 numpy, then cast); the arithmetic of ``forward`` runs in libhelmnet_hip.so
(``hn_laplacian``), which builds its own copy of the constants in C++ -- the GPU tests check
the two against each other.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


def k_grid(n: int) -> np.ndarray:
    """spectral.py:126-127: 2*pi*fftfreq layout with the Nyquist bin kept at -pi."""
    k = 2 * np.pi * np.linspace(-0.5, 0.5, n, endpoint=False)
    return np.concatenate((k[n // 2:], k[: n // 2]))


def pml_profile(n: int, pml: int, sigma_max: float, k: float):
    """1-D sigma, a = -gamma'/gamma^3, b = 1/gamma^2 in float64 (spectral.py:298-363)."""
    i = np.arange(pml)
    outer = sigma_max * (np.abs(1 - i / pml) ** 2)
    sigma = np.zeros(n)
    sigma[:pml] = outer
    sigma[-pml:] = outer[::-1]
    prime = np.zeros(n)
    sp = -2 * sigma_max * (1 - i / pml) / pml
    prime[:pml] = sp
    prime[-pml:] = -sp[::-1]
    inv_gamma = 1.0 / (1.0 + (1j / k) * sigma)
    a = -((1j / k) * prime) * inv_gamma ** 3
    b = inv_gamma ** 2
    return sigma, a, b


def _frozen(t: torch.Tensor) -> nn.Parameter:
    return nn.Parameter(t, requires_grad=False)


class FastLaplacianWithPML(nn.Module):
    def __init__(self, domain_size: int, PMLsize: int, k: float, sigma_max: float):
        super().__init__()
        self._engine = None
        self.init_variables(PMLsize, domain_size, sigma_max, k)

    def init_variables(self, PMLsize, domain_size, sigma_max, k):
        self.PMLsize, self.domain_size, self.sigma_max, self.k = PMLsize, domain_size, sigma_max, k
        n = domain_size
        sigma, a, b = pml_profile(n, PMLsize, sigma_max, k)
        k32 = torch.from_numpy(k_grid(n)).float()          # cast before squaring, as the reference does
        kx = k32.view(1, 1, n, 1).expand(1, n, n, 1)       # varies along the LAST axis ("x" = W)
        ky = k32.view(1, n, 1, 1).expand(1, n, n, 1)
        zero = torch.zeros(1, n, n, 1)
        self.kx = _frozen(torch.cat([zero, kx], -1).contiguous())            # i*kx
        self.ky = _frozen(torch.cat([zero, ky], -1).contiguous())
        self.kx_sq = _frozen(torch.cat([-kx.pow(2), zero], -1).contiguous())  # -kx^2
        self.ky_sq = _frozen(torch.cat([-ky.pow(2), zero], -1).contiguous())

        def pair(c, along_x):
            t = torch.from_numpy(np.stack([c.real, c.imag], -1)).float()     # [n, 2]
            t = t.view(1, 1, n, 2).expand(1, n, n, 2) if along_x else t.view(1, n, 1, 2).expand(1, n, n, 2)
            return _frozen(t.contiguous())

        self.ax, self.bx = pair(a, True), pair(b, True)
        self.ay, self.by = pair(a, False), pair(b, False)
        s = torch.from_numpy(sigma).float()
        self.sigma_x = _frozen(s.view(1, n).expand(n, n).contiguous())
        self.sigma_y = _frozen(s.view(n, 1).expand(n, n).contiguous())

    def sigmas(self):
        return self.sigma_x, self.sigma_y

    def bind(self, engine):
        """Share the owning solver's engine (its hn_set_domain already matches this module)."""
        self._engine = engine

    def _get_engine(self, device):
        from .engine import Engine
        key = (int(self.domain_size), int(self.PMLsize), float(self.sigma_max), float(self.k))
        if self._engine is None or self._engine.device != torch.device(device):
            self._engine = Engine(device)
        if self._engine.domain_key != key:
            self._engine.set_domain(*key)
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, N, N, 2] (re, im last) -> L(x), same layout (spectral.py:251-262)."""
        eng = self._get_engine(x.device)
        nchw = x.permute(0, 3, 1, 2).contiguous()
        return eng.laplacian(nchw).permute(0, 2, 3, 1).contiguous()
