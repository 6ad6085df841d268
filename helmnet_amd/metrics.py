"""Accuracy metrics of the reference's evaluation (host side, plain tensor reductions).

Reference helmnet/support_functions.py:124-130 (normalize_wavefield), :23-48 (difference_to_kwave:
both fields normalised by their value at the source pixel, reference conjugated, PML cropped),
:10-20 (last_frame_difference).  The reference hard-codes the training source pixel [82, 48];
here it is an argument with that default.
"""
from __future__ import annotations

import torch


def as_complex(wf: torch.Tensor) -> torch.Tensor:
    """[B, 2, H, W] (re, im channels) -> complex [B, H, W]."""
    return torch.complex(wf[:, 0].contiguous(), wf[:, 1].contiguous())


def normalize_wavefield(wavefield: torch.Tensor, source_location):
    r, c = source_location
    if wavefield.dim() == 2:
        return wavefield / wavefield[r, c]
    return wavefield / wavefield[:, r, c].unsqueeze(1).unsqueeze(1)


def difference_to_reference(sample: torch.Tensor, reference: torch.Tensor, mask=None, pml_size: int = 10,
                            source_location=(82, 48), conjugate_reference: bool = True):
    """|sample - reference| after source normalisation, cropped by ``pml_size`` (complex [B, H, W] inputs).
    Returns (difference, normalised sample, normalised reference) like the reference's difference_to_kwave."""
    sample = normalize_wavefield(sample, source_location)
    sample = torch.where(torch.isnan(sample.real) | torch.isnan(sample.imag), torch.zeros_like(sample), sample)
    reference = normalize_wavefield(reference, source_location)
    if conjugate_reference:
        reference = torch.conj(reference)
    max_vals = 1
    if mask is not None:
        sample, reference = sample * mask, reference * mask
        max_vals = reference.abs().flatten(1).max(dim=1).values.view(-1, 1, 1)
    diff = torch.abs(sample - reference)[:, pml_size:-pml_size, pml_size:-pml_size] / max_vals
    return diff, sample, reference


def last_frame_difference(stream: torch.Tensor, reference: torch.Tensor, mask=None, **kw):
    """stream [B, T, 2, H, W] of wavefields -> (l_inf [B], rmse [B]) of the last frame against ``reference``."""
    with torch.no_grad():
        last = torch.complex(stream[:, -1, 0], stream[:, -1, 1])
        diff, _, _ = difference_to_reference(last, reference, mask=mask, **kw)
        l_inf = diff.flatten(1).max(dim=1).values
        rmse = diff.pow(2).mean([1, 2]).sqrt()
    return l_inf, rmse
