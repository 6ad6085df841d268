"""Matrix-free restarted GMRES on the SAME HIP Helmholtz operator the learned solver uses.

The reference's classical baseline is MATLAB's ``gmres`` on an explicitly assembled spectral PML
operator (matlab/spectral_gmres_solver.m:50-115).  Here the operator application
A u = L(u) + k_sq * u is libhelmnet_hip.so's fused residual kernel (``hn_residual`` with a zero
source); the Krylov bookkeeping (dot products, axpys, the small Hessenberg least-squares problem) is
ordinary tensor plumbing.  All samples of a batch are solved independently in lock step.
"""
from __future__ import annotations

from typing import Optional

import torch


def _dot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Complex inner product <a, b> = sum conj(a) * b per sample; fields are [B, 2, H, W] (re, im)."""
    re = (a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]).flatten(1).sum(1)
    im = (a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]).flatten(1).sum(1)
    return torch.complex(re.double(), im.double())


def _scale(v: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """v * c with complex per-sample c."""
    cr, ci = c.real.float().view(-1, 1, 1), c.imag.float().view(-1, 1, 1)
    return torch.stack([v[:, 0] * cr - v[:, 1] * ci, v[:, 0] * ci + v[:, 1] * cr], 1)


def gmres(solver, sos_maps: torch.Tensor, restart: int = 20, max_outer: int = 50, tol: float = 1e-4,
          x0: Optional[torch.Tensor] = None):
    """Solve (L + k_sq) u = source for every map of ``sos_maps`` [B, 1, N, N].

    ``solver`` is a helmnet_amd.IterativeSolver on the GPU (its Laplacian tables and source are used).
    Returns dict(wavefield [B,2,N,N], residual_norms list of [B] RMSE after every inner iteration,
    iterations).  Stops when every sample's RMSE (hybridnet.py:295-297 definition) is below ``tol``.
    """
    eng = solver.engine()
    sos_maps = sos_maps.float().contiguous()
    k_sq, wf0 = solver.get_initials(sos_maps)
    k_sq = k_sq.contiguous()
    b_rhs = solver.source.detach().float()
    bsz = sos_maps.shape[0]
    b_rhs = b_rhs.expand(bsz, -1, -1, -1).contiguous() if b_rhs.shape[0] == 1 else b_rhs.contiguous()
    zero_src = torch.zeros_like(b_rhs[:1])
    x = wf0 if x0 is None else x0.float().clone().contiguous()
    npix = float(x[0].numel())

    def apply_a(v):
        return eng.residual(v.contiguous(), k_sq, zero_src)

    history, its = [], 0
    for _ in range(max_outer):
        r = b_rhs - apply_a(x)
        beta = torch.sqrt(_dot(r, r).real)                       # [B]
        history.append((beta / npix ** 0.5).float())
        if float(history[-1].max()) < tol:
            break
        V = [_scale(r, (1.0 / beta.clamp_min(1e-300)).to(torch.complex128))]
        H = torch.zeros(bsz, restart + 1, restart, dtype=torch.complex128, device=x.device)
        g = torch.zeros(bsz, restart + 1, dtype=torch.complex128, device=x.device)
        g[:, 0] = beta
        k_used = 0
        for k in range(restart):
            w = apply_a(V[k])
            for i in range(k + 1):                               # modified Gram-Schmidt
                h = _dot(V[i], w)
                H[:, i, k] = h
                w = w - _scale(V[i], h)
            hn = torch.sqrt(_dot(w, w).real)
            H[:, k + 1, k] = hn
            V.append(_scale(w, (1.0 / hn.clamp_min(1e-300)).to(torch.complex128)))
            k_used = k + 1
            its += 1
            y = torch.linalg.lstsq(H[:, : k + 2, : k + 1], g[:, : k + 2].unsqueeze(-1)).solution.squeeze(-1)
            res = torch.linalg.norm(g[:, : k + 2] - (H[:, : k + 2, : k + 1] @ y.unsqueeze(-1)).squeeze(-1), dim=1)
            history.append((res.real / npix ** 0.5).float())
            if float(history[-1].max()) < tol:
                break
        y = torch.linalg.lstsq(H[:, : k_used + 1, :k_used], g[:, : k_used + 1].unsqueeze(-1)).solution.squeeze(-1)
        for i in range(k_used):
            x = x + _scale(V[i], y[:, i])
    return {"wavefield": x, "residual_norms": history, "iterations": its}
