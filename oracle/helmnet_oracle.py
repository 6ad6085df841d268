"""CPU oracle for the helmnet IterativeSolver inference loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``helmnet_amd/`` (the product) may
import this module.  Its only legitimate users are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- and
there only as the checker / the reported CPU baseline, never as the thing that
is measured or shipped.

It restates, with stock PyTorch CPU ops (fp32, or fp64 on request), the
algorithm of the reference's hot path.  Every function cites the reference
file:line it follows (paths relative to the reference repo root):

  helmnet/spectral.py      k-grid, PML coefficients, fast_laplacian_with_pml
  helmnet/source_module.py point source map
  helmnet/architectures.py DoubleConv / EncoderBlock / HybridNet
  helmnet/hybridnet.py     get_initials / get_residual / single_step / forward

Parity pin: the reference has no tests and no golden vectors of its own
(SURVEY.md section 4).  This oracle is pinned against outputs of the reference
itself, imported unmodified in the build container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks the oracle against every one of them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# Spectral operator tables
# --------------------------------------------------------------------------
def k_grid_1d(n: int) -> np.ndarray:
    """1-D angular wavenumber grid, Nyquist kept at -pi.

    helmnet/spectral.py:126-127 (FourierDerivative.__init__):
    ``2*pi*linspace(-0.5, 0.5, n, endpoint=False)`` rotated by ``n//2``.
    """
    k = 2.0 * np.pi * np.linspace(-0.5, 0.5, n, endpoint=False)
    return np.concatenate((k[n // 2:], k[: n // 2]))


def pml_profiles(n: int, pml: int, sigma_max: float, k: float):
    """1-D PML profiles (float64): sigma, a = -gamma' * invgamma^3, b = invgamma^2.

    helmnet/spectral.py:298-363 (FastLaplacianWithPML.get_gamma_functions).
    The reference builds 2-D meshgrids; the x tables vary only along the last
    axis and the y tables only along the first, so the 1-D profile holds all
    the information (``sigma_x[i, j] = sigma[j]``, ``sigma_y[i, j] = sigma[i]``).
    """
    coord = np.arange(pml)
    sigma_outer = sigma_max * (np.abs(1 - coord / pml) ** 2)  # :306
    sigma = np.zeros((n,))
    sigma[:pml] = sigma_outer
    sigma[-pml:] = np.flip(sigma_outer)
    inv_gamma = 1.0 / (np.ones_like(sigma) + (1j / k) * sigma)  # :317-320
    sp = -2 * sigma_max * (1 - coord / pml) / pml  # :323-325
    sigma_prime = np.zeros((n,))
    sigma_prime[:pml] = sp
    sigma_prime[-pml:] = -np.flip(sp)
    gamma_prime = (1j / k) * sigma_prime  # :330
    a = -gamma_prime * inv_gamma ** 3  # :335
    b = inv_gamma ** 2  # :336
    return sigma, a, b


class SpectralTables:
    """All constants of FastLaplacianWithPML for one domain size.

    helmnet/spectral.py:267-296 (init_variables) and :298-363.  Layout
    matches the reference: ``[1, N, N, 2]`` (re, im) float32 tensors, plus the
    two ``[N, N]`` sigma maps (hybridnet.py:126-131 stacks them as ``sigmas``).
    """

    def __init__(self, n: int, pml: int, sigma_max: float, k: float, dtype=torch.float32):
        self.n = n
        k1 = k_grid_1d(n)
        sigma, a, b = pml_profiles(n, pml, sigma_max, k)
        kx2d, ky2d = np.meshgrid(k1, k1)  # kx[i, j] = k1[j]; ky[i, j] = k1[i]  (:130-139)
        # the reference casts k to float32 *before* squaring (:274-277)
        kx32 = torch.from_numpy(kx2d).float()
        ky32 = torch.from_numpy(ky2d).float()
        z = torch.zeros_like(kx32)

        def ri(re, im):
            return torch.stack([re, im], dim=-1).unsqueeze(0).to(dtype)

        self.kx = ri(z, kx32)  # i*kx  (:281)
        self.ky = ri(z, ky32)
        self.kx_sq = ri(-(kx32 ** 2), z)  # -kx^2 (:283)
        self.ky_sq = ri(-(ky32 ** 2), z)
        sx, sy = np.meshgrid(sigma, sigma)
        ax2, _ = np.meshgrid(a, a)
        bx2, _ = np.meshgrid(b, b)
        ay2 = ax2.T
        by2 = bx2.T

        def c2t(c):
            return ri(torch.from_numpy(np.real(c).copy()).float(), torch.from_numpy(np.imag(c).copy()).float())

        self.ax, self.bx, self.ay, self.by = c2t(ax2), c2t(bx2), c2t(ay2), c2t(by2)
        self.sigma_x = torch.from_numpy(sx).float().to(dtype)
        self.sigma_y = torch.from_numpy(sy).float().to(dtype)
        self.sigmas = torch.stack([self.sigma_x, self.sigma_y], 0)  # hybridnet.py:126-131


def complex_mul(x: Tensor, y: Tensor) -> Tensor:
    """helmnet/spectral.py:6-18 -- (re, im) pairs in the last dimension."""
    re = x[..., 0] * y[..., 0] - x[..., 1] * y[..., 1]
    im = x[..., 1] * y[..., 0] + x[..., 0] * y[..., 1]
    return torch.stack([re, im], dim=-1)


def laplacian_nhwc(u: Tensor, t: SpectralTables) -> Tensor:
    """helmnet/spectral.py:31-79 (fast_laplacian_with_pml), ``u`` is [B,N,N,2].

    One forward 2-D c2c FFT, four spectral multiplies, four inverse 2-D FFTs,
    then the spatially varying PML coefficients.
    """
    uf = torch.view_as_real(torch.fft.fftn(torch.view_as_complex(u.contiguous()), dim=(-2, -1), norm="backward"))
    stack = torch.stack(
        [complex_mul(uf, t.kx), complex_mul(uf, t.ky), complex_mul(uf, t.kx_sq), complex_mul(uf, t.ky_sq)], dim=0
    )
    d = torch.view_as_real(torch.fft.ifftn(torch.view_as_complex(stack), dim=(-2, -1), norm="backward"))
    return complex_mul(t.ax, d[0]) + complex_mul(t.ay, d[1]) + complex_mul(t.bx, d[2]) + complex_mul(t.by, d[3])


def apply_laplacian(x: Tensor, t: SpectralTables) -> Tensor:
    """helmnet/hybridnet.py:540-542 -- NCHW in, NCHW out."""
    return laplacian_nhwc(x.permute(0, 2, 3, 1).contiguous(), t).permute(0, 3, 1, 2)


def get_residual(wf: Tensor, k_sq: Tensor, source: Tensor, t: SpectralTables) -> Tensor:
    """helmnet/hybridnet.py:544-556: L(wf) + k_sq * wf - source."""
    return apply_laplacian(wf, t) + k_sq * wf - source


def get_initials(sos: Tensor, omega: float) -> Tuple[Tensor, Tensor]:
    """helmnet/hybridnet.py:522-538: k_sq = (omega/sos)^2, wavefield = zeros (fp32)."""
    k_sq = (omega / sos) ** 2
    wf = torch.zeros(k_sq.shape[0], 2, k_sq.shape[2], k_sq.shape[3], dtype=k_sq.dtype)
    return k_sq, wf


def test_loss_function(res: Tensor) -> Tensor:
    """helmnet/hybridnet.py:295-297: per-sample residual RMSE."""
    return res.pow(2).mean((1, 2, 3)).sqrt()


def axis_operator_matrix(n: int, pml: int, sigma_max: float, k: float) -> np.ndarray:
    """Explicit complex128 [n, n] matrix of the 1-D operator  a . IDFT diag(i k) DFT + b . IDFT diag(-k^2) DFT  the
    spectral Laplacian applies along one axis, assembled from DFT matrices the way the reference's classical baseline
    assembles its derivative operators (matlab/spectral_gmres_solver.m:50-82: ``dftx = fft(eye(Nx))``, ``Dx = idftx *
    (ikx * dftx)``), with the fp32-rounded k / a / b tables of spectral.py (so it is the same operator, in float64)."""
    k1 = k_grid_1d(n).astype(np.float32)
    k2 = -(k1 * k1)                                       # fp32 square of the fp32 grid (spectral.py:283)
    _, a, b = pml_profiles(n, pml, sigma_max, k)
    a = a.real.astype(np.float32).astype(np.float64) + 1j * a.imag.astype(np.float32).astype(np.float64)
    b = b.real.astype(np.float32).astype(np.float64) + 1j * b.imag.astype(np.float32).astype(np.float64)
    dft = np.fft.fft(np.eye(n))
    idft = np.fft.ifft(np.eye(n))
    d1 = idft @ (np.diag(1j * k1.astype(np.float64)) @ dft)
    d2 = idft @ (np.diag(k2.astype(np.float64)) @ dft)
    return np.diag(a) @ d1 + np.diag(b) @ d2


def assemble_helmholtz_matrix(k_sq: np.ndarray, pml: int, sigma_max: float, k: float) -> np.ndarray:
    """Full system matrix M = A + B + D of matlab/spectral_gmres_solver.m:77-90 for ONE [n, n] map of k_sq, acting on
    the row-major flattened complex wavefield: A = I (x) Mx (along W), B = My (x) I (along H), D = diag(k_sq)."""
    n = k_sq.shape[-1]
    m1 = axis_operator_matrix(n, pml, sigma_max, k)
    eye = np.eye(n)
    return np.kron(eye, m1) + np.kron(m1, eye) + np.diag(k_sq.reshape(-1).astype(np.float64))


def direct_solve(sos: np.ndarray, source: np.ndarray, pml: int, sigma_max: float, k: float, omega: float = 1.0) -> np.ndarray:
    """float64 direct solution of (L + k_sq) u = source for one [n, n] sound-speed map; source [2, n, n] (re, im)
    -> [2, n, n].  The role MATLAB's gmres to 1e-10 plays in the reference (spectral_gmres_solver.m:92-107)."""
    k_sq = (omega / sos.astype(np.float64)) ** 2
    mat = assemble_helmholtz_matrix(k_sq, pml, sigma_max, k)
    rhs = (source[0].astype(np.float64) + 1j * source[1].astype(np.float64)).reshape(-1)
    u = np.linalg.solve(mat, rhs).reshape(k_sq.shape)
    return np.stack([u.real, u.imag])


# --------------------------------------------------------------------------
# Source
# --------------------------------------------------------------------------
def point_source_map(n: int, location: Sequence[int], amplitude: float, omega: float = 1.0,
                     phase: float = 0.0, smooth: bool = False, t: float = 0.0) -> Tensor:
    """helmnet/source_module.py:41-79,94-116 + hybridnet.py:151-153 -> [1,2,N,N].

    |ifft2(ifftshift(fftshift(fft2(delta*amp)) [* blackman^2]))| times
    (cos, sin)(omega*t + phase).
    """
    m = torch.zeros((n, n))
    m[location[0], location[1]] = amplitude
    f = torch.fft.fftshift(torch.fft.fft2(m))
    if smooth:
        w = torch.blackman_window(n)
        f = f * torch.outer(w, w)
    a = torch.abs(torch.fft.ifft2(torch.fft.ifftshift(f)))
    ct = torch.tensor(omega * t + phase)
    src = torch.stack([a * torch.cos(ct), a * torch.sin(ct)], dim=2).unsqueeze(0)
    return src.permute(0, 3, 1, 2)


# --------------------------------------------------------------------------
# UNet (HybridNet)
# --------------------------------------------------------------------------
def prelu(x: Tensor, a: Tensor) -> Tensor:
    """helmnet/architectures.py:32-33 -- nn.PReLU() with ONE scalar slope."""
    return torch.clamp_min(x, 0) + a * torch.clamp_max(x, 0)


def activation(h: Tensor, name: str, slope: Optional[Tensor]) -> Tensor:
    """helmnet/architectures.py:5-44 (getActivationFunction), default arguments of the torch modules it returns."""
    name = name.lower()
    if name == "prelu":
        return prelu(h, slope)
    if name == "relu":
        return torch.clamp_min(h, 0)
    if name == "leakyrelu":
        return torch.clamp_min(h, 0) + 0.01 * torch.clamp_max(h, 0)
    if name == "celu":
        return torch.clamp_min(h, 0) + torch.clamp_max(torch.expm1(h), 0)
    if name == "tanh":
        return torch.tanh(h)
    if name == "gelu":
        return 0.5 * h * (1.0 + torch.erf(h * 0.7071067811865476))
    if name == "tanhshrink":
        return h - torch.tanh(h)
    if name == "softplus":
        return torch.where(h > 20, h, torch.log1p(torch.exp(h)))
    raise NotImplementedError(name)


def _keep(tape: Optional[dict], name: str, x: Tensor) -> Tensor:
    """Record an intermediate tensor (and ask autograd to keep its gradient) -- used by the training parity tests to
    localise a mismatch; no effect when ``tape`` is None."""
    if tape is not None:
        force = tape.get("__force__")
        if force is not None and name in force:
            # evaluate the graph AT the given value (straight-through: the gradient passes unchanged).  The HIP path's
            # pre-activations differ from these by fp32 rounding; a PReLU input within rounding of zero then takes the other
            # branch, and a comparison of gradients would measure that coin flip instead of the kernels
            x = force[name].to(x.dtype).detach() + (x - x.detach())   # value: exactly the forced one; gradient: identity
        if x.requires_grad:
            x.retain_grad()
        tape[name] = x
    return x


def double_conv(x: Tensor, w: Dict[str, Tensor], prefix: str, act: str = "prelu", tape: Optional[dict] = None) -> Tensor:
    """helmnet/architectures.py:63-84: conv3x3(pad 1) -> activation -> conv3x3(pad 1)."""
    p = prefix + ".double_conv."
    h = F.conv2d(x, w[p + "0.weight"], w[p + "0.bias"], padding=1)
    h = _keep(tape, prefix + ".mid", h)
    h = activation(h, act, w.get(p + "1.weight"))
    return F.conv2d(h, w[p + "2.weight"], w[p + "2.bias"], padding=1)


def state_dims(n: int, depth: int) -> List[int]:
    """helmnet/architectures.py:390-392."""
    return [n // 2 ** d for d in range(depth)]


def flatten_states(states: List[Tensor]) -> Tensor:
    """helmnet/architectures.py:425-429."""
    return torch.cat([s.reshape(s.shape[0], s.shape[1], -1) for s in states], 2)


def unflatten_states(flat: Tensor, n: int, depth: int) -> List[Tensor]:
    """helmnet/architectures.py:431-437."""
    out, o = [], 0
    for s in state_dims(n, depth):
        out.append(flat[:, :, o:o + s * s].reshape(flat.shape[0], flat.shape[1], s, s))
        o += s * s
    return out


def unet_forward(x6: Tensor, states: List[Tensor], w: Dict[str, Tensor], depth: int = 4, act: str = "prelu",
                 state_depth: int = None, tape: Optional[dict] = None) -> Tuple[Tensor, List[Tensor]]:
    """helmnet/architectures.py:439-465 (HybridNet.forward) with
    EncoderBlock.forward (:240-252) inlined; levels d >= state_depth run without state (:250-251) and keep
    whatever their state slot held.  Returns (d, new_states)."""
    state_depth = depth if state_depth is None else state_depth
    x = _keep(tape, "x0", double_conv(x6, w, "inc", act, tape))
    skips, new_states = [], []
    for d in range(depth):
        if d >= state_depth:
            out = double_conv(x, w, f"enc.{d}.conv_signal", act)
            new_states.append(states[d])
            skips.append(out)
            x = F.conv2d(out, w[f"enc.{d}.down.weight"], w[f"enc.{d}.down.bias"], stride=2, padding=3)
            continue
        out = _keep(tape, f"out{d}", double_conv(torch.cat([x, states[d]], 1), w, f"enc.{d}.conv_signal", act, tape))
        new_states.append(double_conv(torch.cat([out, states[d]], 1), w, f"enc.{d}.conv_state", act, tape))
        skips.append(out)
        x = _keep(tape, f"x{d + 1}", F.conv2d(out, w[f"enc.{d}.down.weight"], w[f"enc.{d}.down.bias"], stride=2, padding=3))
    x = _keep(tape, f"y{depth}", double_conv(x, w, f"decode.{depth}", act, tape))
    for d in range(depth - 1, -1, -1):
        x = _keep(tape, f"u{d}", F.conv_transpose2d(x, w[f"up.{d}.weight"], w[f"up.{d}.bias"], stride=2, padding=3))
        x = _keep(tape, f"y{d}", double_conv(torch.cat([x, skips[d]], 1), w, f"decode.{d}", act, tape))
    return F.conv2d(x, w["outc.conv.weight"], w["outc.conv.bias"]), new_states


# --------------------------------------------------------------------------
# Solver loop
# --------------------------------------------------------------------------
def single_step(wf, k_sq, res, states, w, source, t: SpectralTables, depth: int = 4, act: str = "prelu", tape: Optional[dict] = None,
                state_depth: Optional[int] = None):
    """helmnet/hybridnet.py:558-584."""
    sig = t.sigmas.to(wf.dtype).unsqueeze(0).repeat(wf.shape[0], 1, 1, 1)
    inp = torch.cat([wf, 1e3 * res, sig], dim=1)
    d, new_states = unet_forward(inp, states, w, depth, act, state_depth=state_depth, tape=tape)
    up = d / 1e3 + wf
    return up, get_residual(up, k_sq, source, t), new_states


def solve(sos: Tensor, w: Dict[str, Tensor], source: Tensor, t: SpectralTables, num_iterations: int,
          omega: float = 1.0, depth: int = 4, keep: str = "rmse"):
    """helmnet/hybridnet.py:654-697 (IterativeSolver.forward).

    keep = "rmse" returns the per-iteration per-sample RMSE trace instead of
    the full residual list (the reference keeps every residual tensor).
    """
    k_sq, wf = get_initials(sos, omega)
    n = sos.shape[-1]
    states = [torch.zeros(sos.shape[0], 2, s, s, dtype=sos.dtype) for s in state_dims(n, depth)]
    res = get_residual(wf, k_sq, source, t)
    trace = []
    for _ in range(num_iterations):
        wf, res, states = single_step(wf, k_sq, res, states, w, source, t, depth)
        trace.append(test_loss_function(res) if keep == "rmse" else res)
    return {"wavefield": wf, "residual": res, "states": states, "trace": trace}


# --------------------------------------------------------------------------
# Training step
# --------------------------------------------------------------------------
def training_loss(wf: Tensor, res: Tensor, states_flat: Tensor, k_sq: Tensor, source: Tensor, w: Dict[str, Tensor],
                  t: SpectralTables, n_unroll: int = 10, depth: int = 4, act: str = "prelu", loss_scale: float = 1e4,
                  tape: Optional[dict] = None, state_depth: Optional[int] = None):
    """helmnet/hybridnet.py:399-409 (the differentiable core of ``training_step``): ``f.set_states(h_states, flatten=True)``
    -> ``n_steps(wavefields, k_sqs, residual, unrolling_steps, True, True)`` (:586-623) -> ``loss = 1e4 * cat(residuals).pow(2).mean()``.
    Every argument may require grad; gradients come from ``torch.autograd`` exactly as in the reference.  ``tape`` (optional
    dict) receives the intermediate tensors of the FIRST unrolled iteration.  Returns (loss, wavefields, residuals, flat states)."""
    n = wf.shape[-1]
    states = unflatten_states(states_flat, n, depth)
    wfs, ress, sts = [], [], []
    for it in range(n_unroll):
        wf, res, states = single_step(wf, k_sq, res, states, w, source, t, depth, act, tape if it == 0 else None, state_depth)
        wfs.append(wf)
        ress.append(res)
        sts.append(flatten_states(states))
    loss = loss_scale * torch.cat(ress).pow(2).mean()
    return loss, wfs, ress, sts


def adam_reference(weights: Tensor, grads: Sequence[Tensor], lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                   weight_decay: float = 0.0, clip_value: float = 0.0) -> Tensor:
    """helmnet/hybridnet.py:172-176 + :250-258 on ONE flat parameter vector: for every gradient in ``grads`` (one optimiser step
    each) ``clip_grad_value_`` then ``torch.optim.Adam(lr, betas, weight_decay).step()``.  Returns the updated copy."""
    p = torch.nn.Parameter(weights.clone())
    opt = torch.optim.Adam([p], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    for g in grads:
        opt.zero_grad()
        p.grad = g.clone()
        if clip_value > 0:
            torch.nn.utils.clip_grad_value_([p], clip_value)
        opt.step()
    return p.detach()
