"""Host-side mirror of the reference's monochromatic source module.

Mirrors ``SourceModule`` (reference helmnet/source_module.py:4-116): a point source of given
amplitude, optionally smoothed with a Blackman window in the spatial-frequency domain,
multiplied by (cos, sin)(omega*t + phase).  This runs once per ``set_domain_size`` (setup, not
the per-iteration loop), so it is evaluated with stock fp32 torch ops on the host and moved to
the module's device.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class SourceModule(nn.Module):
    def __init__(self, image_size: int, omega: float = 1, location=(180, 50), amplitude: float = 1.0,
                 phase: float = 0.0, smooth: bool = True):
        super().__init__()
        self.L = image_size
        self.location = list(location)
        self.t = None
        self.omega = omega
        self.amplitude = amplitude
        self.phase = phase
        self.smooth = smooth
        self.register_buffer("_dummy_for_device", torch.tensor(1))
        self.register_buffer("_abs_spatial_map", None)
        self.make_abs_spatial_map(smooth=smooth)

    def make_abs_spatial_map(self, smooth: bool = True):
        """source_module.py:41-79: |ifft2(ifftshift(fftshift(fft2(delta)) [* blackman^2]))|."""
        m = torch.zeros((self.L, self.L))
        m[self.location[0], self.location[1]] = self.amplitude
        f = torch.fft.fftshift(torch.fft.fft2(m))
        if smooth:
            w = torch.blackman_window(self.L)
            f = f * torch.outer(w, w)
        self._abs_spatial_map = torch.abs(torch.fft.ifft2(torch.fft.ifftshift(f))).to(self._dummy_for_device.device)

    def set_new_location(self, location):
        if not (self.location[0] == location[0] and self.location[1] == location[1]):
            self.location = list(location)
            self.t = None
            self.make_abs_spatial_map(smooth=self.smooth)

    def get_location(self):
        return self.location

    def spatial_map(self, t: float) -> torch.Tensor:
        """source_module.py:94-116 -> [1, L, L, 2]."""
        ang = float(self.omega * t + self.phase)
        a = self._abs_spatial_map
        return torch.stack([a * math.cos(ang), a * math.sin(ang)], dim=2).unsqueeze(0)
