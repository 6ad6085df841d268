"""Thin tensor-level wrapper over one ``hn_ctx`` of libhelmnet_hip.so.

PyTorch is used for device memory and streams only: every method checks its tensors
(device, dtype, contiguity, shape), passes raw device pointers plus the current HIP stream
across the C ABI, and returns.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

# state_dict order of the `f.*` tensors = order of the blob hn_load_weights expects
# (reference helmnet/architectures.py:317-388)


def weight_names(depth: int = 4) -> list:
    def dc(prefix):
        p = prefix + ".double_conv."
        return [p + "0.weight", p + "0.bias", p + "1.weight", p + "2.weight", p + "2.bias"]

    names = dc("inc")
    for d in range(depth):
        names += dc(f"enc.{d}.conv_signal") + [f"enc.{d}.down.weight", f"enc.{d}.down.bias"] + dc(f"enc.{d}.conv_state")
    for d in range(depth + 1):
        names += dc(f"decode.{d}")
    for d in range(depth):
        names += [f"up.{d}.weight", f"up.{d}.bias"]
    return names + ["outc.conv.weight", "outc.conv.bias"]


def pack_weights(state: dict, depth: int = 4, activation: str = "prelu", state_depth: Optional[int] = None) -> np.ndarray:
    """Flatten a HybridNet state_dict (tensors or arrays) into the fp32 blob of hn_load_weights.
    Parameter-free activations (relu / leakyrelu) get their constant slope written where the
    PReLU weight would be, so the blob layout never changes.

    ``state_depth < depth`` (architectures.py:353, ``use_state = d < state_depth``): the kernels always run the stateful
    layout, so an encoder level WITHOUT state is packed as its exact stateful equivalent -- conv_signal's first
    convolution [8, 8, 3, 3] gets two all-zero input channels where the state would be concatenated (their products are
    exact zeros), and the missing conv_state becomes an all-zero DoubleConv; the host leaves that level's state slice
    untouched (HybridNet.adopt_states)."""
    const_slope = {"relu": 0.0, "leakyrelu": 0.01, "celu": 0.0, "tanh": 0.0, "gelu": 0.0, "tanhshrink": 0.0, "softplus": 0.0}
    state_depth = depth if state_depth is None else state_depth
    stateless = {f"enc.{d}." for d in range(state_depth, depth)}
    zero_state_dc = {"0.weight": (2, 10, 3, 3), "0.bias": (2,), "2.weight": (2, 2, 3, 3), "2.bias": (2,)}
    parts = []
    for name in weight_names(depth):
        level = name[: name.index(".", 4) + 1] if name.startswith("enc.") else None
        if level in stateless and ".conv_state." in name:
            tail = name.split(".double_conv.")[1]
            parts.append(np.full(1, 0.25, np.float32) if tail == "1.weight" else np.zeros(int(np.prod(zero_state_dc[tail])), np.float32))
            continue
        if name.endswith(".double_conv.1.weight") and name not in state:
            if activation.lower() not in const_slope:
                raise KeyError(f"missing {name} for activation {activation!r}")
            parts.append(np.array([const_slope[activation.lower()]], np.float32))
            continue
        v = state[name]
        if isinstance(v, torch.Tensor):
            v = v.detach().to("cpu", torch.float32).numpy()
        v = np.ascontiguousarray(v, dtype=np.float32)
        if level in stateless and name.endswith(".conv_signal.double_conv.0.weight"):
            assert v.shape[1] == 8, v.shape
            v = np.concatenate([v, np.zeros((v.shape[0], 2, 3, 3), np.float32)], 1)
        parts.append(v.reshape(-1))
    return np.concatenate(parts)


def weight_shapes(depth: int = 4, features: int = 8, state_channels: int = 2, inchannels: int = 6) -> dict:
    """name -> shape of every tensor of the blob, in blob order (architectures.py:63-84, 209-211, 340-388)."""
    f, s = features, state_channels

    def dc(prefix, cin, cm, co):
        p = prefix + ".double_conv."
        return {p + "0.weight": (cm, cin, 3, 3), p + "0.bias": (cm,), p + "1.weight": (1,), p + "2.weight": (co, cm, 3, 3), p + "2.bias": (co,)}

    shapes = dc("inc", inchannels, f, f)
    for d in range(depth):
        shapes.update(dc(f"enc.{d}.conv_signal", f + s, f, f))
        shapes.update({f"enc.{d}.down.weight": (f, f, 8, 8), f"enc.{d}.down.bias": (f,)})
        shapes.update(dc(f"enc.{d}.conv_state", f + s, s, s))
    for d in range(depth + 1):
        shapes.update(dc(f"decode.{d}", 2 * f if d < depth else f, f, f))
    for d in range(depth):
        shapes.update({f"up.{d}.weight": (f, f, 8, 8), f"up.{d}.bias": (f,)})
    shapes.update({"outc.conv.weight": (2, f, 1, 1), "outc.conv.bias": (2,)})
    assert list(shapes) == weight_names(depth)
    return shapes


def unpack_weights(blob, depth: int = 4) -> dict:
    """Inverse of pack_weights for state_depth == depth: flat fp32 blob (array or tensor) -> {name: array in PyTorch layout}."""
    if isinstance(blob, torch.Tensor):
        blob = blob.detach().to("cpu", torch.float32).numpy()
    out, pos = {}, 0
    for name, shape in weight_shapes(depth).items():
        n = int(np.prod(shape))
        out[name] = np.array(blob[pos:pos + n], np.float32).reshape(shape)
        pos += n
    assert pos == blob.size, (pos, blob.size)
    return out


_MODULE_ENGINES: dict = {}


def module_engine(device) -> "Engine":
    """One shared context per device for the standalone sub-module forwards (DoubleConv / OutConv / EncoderBlock called directly)."""
    dev = torch.device(device)
    dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device()) if dev.type == "cuda" else dev
    if dev not in _MODULE_ENGINES:
        _MODULE_ENGINES[dev] = Engine(dev)
    return _MODULE_ENGINES[dev]


def release_module_engines():
    """Destroy the shared per-device contexts of the standalone sub-module forwards (they otherwise live until interpreter exit)."""
    for eng in _MODULE_ENGINES.values():
        eng.close()
    _MODULE_ENGINES.clear()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Engine:
    """One library context bound to one HIP device."""

    def __init__(self, device: torch.device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.HelmnetHipError(
                f"helmnet_amd computes on MI355X (device type 'cuda' under ROCm) only, got {device}; "
                "there is no CPU path in the product (the CPU oracle lives in oracle/ for tests)."
            )
        self.lib = _lib.load()
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        ctx = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream()  # make sure torch initialised HIP on this device first
            _lib.check(self.lib.hn_create(ctypes.byref(ctx), self.device.index), None, "hn_create")
        self.ctx = ctx
        self.n = 0
        self.depth = 0
        self.weights_key = None
        self.domain_key = None

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.hn_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup -------------------------------------------------------------------------
    def load_weights(self, blob: np.ndarray, features: int, depth: int, state_channels: int, activation: str):
        act = _lib.HN_ACT.get(activation.lower())
        if act is None:
            raise NotImplementedError(f"Unknown activation function {activation} (HIP kernels: {sorted(_lib.HN_ACT)})")
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        rc = self.lib.hn_load_weights(self.ctx, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), blob.size,
                                      features, depth, state_channels, act)
        _lib.check(rc, self.ctx, "hn_load_weights")
        self.depth = depth

    def set_domain(self, n: int, pml: int, sigma_max: float, k: float):
        _lib.check(self.lib.hn_set_domain(self.ctx, int(n), int(pml), float(sigma_max), float(k)), self.ctx, "hn_set_domain")
        self.n = int(n)
        self.domain_key = (int(n), int(pml), float(sigma_max), float(k))

    def set_unet_precision(self, mode: str):
        """Arithmetic of the UNet convolutions for this context: 'fp32' (default), 'bf16x3', 'fp16', 'bf16x2', 'valu'."""
        if mode not in _lib.HN_PRECISION:
            raise ValueError(f"unknown UNet precision {mode!r} (choose from {sorted(_lib.HN_PRECISION)})")
        _lib.check(self.lib.hn_set_unet_precision(self.ctx, _lib.HN_PRECISION[mode]), self.ctx, "hn_set_unet_precision")

    @property
    def unet_precision(self) -> str:
        code = self.lib.hn_get_unet_precision(self.ctx)
        names = {v: k for k, v in _lib.HN_PRECISION.items()}
        if code not in names:
            raise _lib.HelmnetHipError(f"hn_get_unet_precision returned {code} (context closed?)")
        return names[code]

    def set_option(self, name: str, value: int):
        """Library tuning knobs (enum hn_option of include/helmnet_hip.h): 'lanes' 1..8, 'side_stream' 0..3, 'deep' 0/1, 'spectral_pfa' 0/1,
        'spectral_radix16' 0..2, 'dc_valu' 0..4, 'spectral_cols' 0..2, 'train_fused', 'train_overlap'; laboratory knobs (HN_EXP_*): 'graph' 0, 1 or an even
        number <= 64 of iterations per graph, 'train_lanes' 1/2 (hn_train_grad: the halves of the batch as two chains of launches on two streams).
        'spectral_pfa' is read when the spectral tables are built: changing it re-builds them."""
        if name not in _lib.HN_OPTION:
            raise ValueError(f"unknown option {name!r} (choose from {sorted(_lib.HN_OPTION)})")
        _lib.check(self.lib.hn_set_option(self.ctx, _lib.HN_OPTION[name], int(value)), self.ctx, "hn_set_option")
        if name == "spectral_pfa" and self.domain_key is not None:
            self.set_domain(*self.domain_key)

    def counter(self, name: str) -> int:
        return int(self.lib.hn_get_counter(self.ctx, _lib.HN_COUNTER[name]))

    def check_async_errors(self):
        """Raise if a bounded device-side wait of any earlier call on this context gave up (hn_check_async_errors); reliable once the stream the work
        ran on has been synchronised."""
        _lib.check(self.lib.hn_check_async_errors(self.ctx), self.ctx, "hn_check_async_errors")

    @property
    def state_len(self) -> int:
        return int(self.lib.hn_state_len(self.ctx))

    def reserve(self, batch: int):
        _lib.check(self.lib.hn_reserve(self.ctx, int(batch)), self.ctx, "hn_reserve")

    # ---- helpers -----------------------------------------------------------------------
    def _chk(self, t: torch.Tensor, shape: Sequence[int], name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"{name} must be a tensor")
        if t.device != self.device:
            raise ValueError(f"{name} is on {t.device}, engine is on {self.device}")
        if t.dtype != torch.float32:
            raise TypeError(f"{name} must be float32, got {t.dtype}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
        return t

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- operators ---------------------------------------------------------------------
    def sigmas(self) -> torch.Tensor:
        out = torch.empty(2, self.n, self.n, device=self.device, dtype=torch.float32)
        _lib.check(self.lib.hn_get_sigmas(self.ctx, _ptr(out), self._stream()), self.ctx, "hn_get_sigmas")
        return out

    def laplacian(self, wf: torch.Tensor) -> torch.Tensor:
        b = wf.shape[0]
        self._chk(wf, (b, 2, self.n, self.n), "wavefield")
        out = torch.empty_like(wf)
        _lib.check(self.lib.hn_laplacian(self.ctx, _ptr(wf), _ptr(out), b, self._stream()), self.ctx, "hn_laplacian")
        return out

    def residual(self, wf: torch.Tensor, k_sq: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
        b = wf.shape[0]
        self._chk(wf, (b, 2, self.n, self.n), "wavefield")
        self._chk(k_sq, (b, 1, self.n, self.n), "k_sq")
        self._chk(src, (src.shape[0], 2, self.n, self.n), "source")
        out = torch.empty_like(wf)
        rc = self.lib.hn_residual(self.ctx, _ptr(wf), _ptr(k_sq), _ptr(src), src.shape[0], _ptr(out), b, self._stream())
        _lib.check(rc, self.ctx, "hn_residual")
        return out

    def residual_vjp(self, g: torch.Tensor, k_sq: torch.Tensor) -> torch.Tensor:
        """J^T g of ``residual`` with respect to the wavefield: L^H(g) + k_sq * g."""
        b = g.shape[0]
        self._chk(g, (b, 2, self.n, self.n), "cotangent")
        self._chk(k_sq, (b, 1, self.n, self.n), "k_sq")
        out = torch.empty_like(g)
        _lib.check(self.lib.hn_residual_vjp(self.ctx, _ptr(g), _ptr(k_sq), _ptr(out), b, self._stream()), self.ctx, "hn_residual_vjp")
        return out

    def rmse(self, res: torch.Tensor) -> torch.Tensor:
        b = res.shape[0]
        self._chk(res, (b, 2, self.n, self.n), "residual")
        out = torch.empty(b, device=self.device, dtype=torch.float32)
        _lib.check(self.lib.hn_rmse(self.ctx, _ptr(res), _ptr(out), b, self._stream()), self.ctx, "hn_rmse")
        return out

    def unet(self, in6: torch.Tensor, states_in: torch.Tensor):
        b = in6.shape[0]
        self._chk(in6, (b, 6, self.n, self.n), "network input")
        self._chk(states_in, (b, 2, self.state_len), "hidden state")
        d = torch.empty(b, 2, self.n, self.n, device=self.device, dtype=torch.float32)
        states_out = torch.empty_like(states_in)
        rc = self.lib.hn_unet(self.ctx, _ptr(in6), _ptr(states_in), _ptr(states_out), _ptr(d), b, self._stream())
        _lib.check(rc, self.ctx, "hn_unet")
        return d, states_out

    # ---- standalone sub-modules (hn_double_conv / hn_conv8x8 / hn_out_conv; utility paths, weights re-packed per call) ----
    @staticmethod
    def _host_blob(parts) -> np.ndarray:
        return np.ascontiguousarray(np.concatenate([np.asarray(p.detach().cpu() if isinstance(p, torch.Tensor) else p, np.float32).reshape(-1)
                                                    for p in parts]))

    def _plain(self, x: torch.Tensor, channels: int, name: str) -> torch.Tensor:
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != channels:
            raise ValueError(f"{name}: expected a [B, {channels}, H, W] tensor, got {tuple(getattr(x, 'shape', ()))}")
        if x.device != self.device:
            raise ValueError(f"{name}: tensor on {x.device}, engine on {self.device}")
        return x.float().contiguous()

    def double_conv(self, x, w1, b1, slope, w2, b2, activation: str = "prelu") -> torch.Tensor:
        """DoubleConv.forward (architectures.py:83-84) for the channel shapes of the UNet."""
        cout, cin = int(w1.shape[0]), int(w1.shape[1])
        x = self._plain(x, cin, "double_conv")
        blob = self._host_blob([w1, b1, np.float32([slope]) if np.isscalar(slope) else slope, w2, b2])
        out = torch.empty(x.shape[0], cout, x.shape[2], x.shape[3], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            rc = self.lib.hn_double_conv(self.ctx, _ptr(x), cin, cout, blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                         _lib.HN_ACT[activation.lower()], _ptr(out), x.shape[0], x.shape[2], x.shape[3], self._stream())
        _lib.check(rc, self.ctx, "hn_double_conv")
        return out

    def conv8x8(self, x, weight, bias, transposed: bool) -> torch.Tensor:
        """nn.Conv2d(8, 8, 8, stride=2, padding=3) / nn.ConvTranspose2d(8, 8, 8, stride=2, padding=3) of the UNet."""
        x = self._plain(x, 8, "conv8x8")
        b, _, h, w = x.shape
        out = torch.empty(b, 8, 2 * h if transposed else h // 2, 2 * w if transposed else w // 2, device=self.device, dtype=torch.float32)
        blob = self._host_blob([weight, bias])
        with torch.cuda.device(self.device):
            rc = self.lib.hn_conv8x8(self.ctx, _ptr(x), blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(bool(transposed)), _ptr(out),
                                     b, h, w, self._stream())
        _lib.check(rc, self.ctx, "hn_conv8x8")
        return out

    def out_conv(self, x, weight, bias) -> torch.Tensor:
        """OutConv.forward (architectures.py:57-60): Conv2d(8, 2, 1)."""
        x = self._plain(x, 8, "out_conv")
        out = torch.empty(x.shape[0], 2, x.shape[2], x.shape[3], device=self.device, dtype=torch.float32)
        blob = self._host_blob([weight, bias])
        with torch.cuda.device(self.device):
            rc = self.lib.hn_out_conv(self.ctx, _ptr(x), blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _ptr(out), x.shape[0], x.shape[2],
                                      x.shape[3], self._stream())
        _lib.check(rc, self.ctx, "hn_out_conv")
        return out

    def step(self, wf, res, states, k_sq, src, n_iter: int, res_hist=None, wf_hist=None, st_hist=None, rmse_hist=None):
        """n_iter solver iterations; wf, res, states updated in place."""
        b = wf.shape[0]
        self._chk(wf, (b, 2, self.n, self.n), "wavefield")
        self._chk(res, (b, 2, self.n, self.n), "residual")
        self._chk(states, (b, 2, self.state_len), "hidden state")
        self._chk(k_sq, (b, 1, self.n, self.n), "k_sq")
        self._chk(src, (src.shape[0], 2, self.n, self.n), "source")
        if res_hist is not None:
            self._chk(res_hist, (n_iter, b, 2, self.n, self.n), "res_hist")
        if wf_hist is not None:
            self._chk(wf_hist, (n_iter, b, 2, self.n, self.n), "wf_hist")
        if st_hist is not None:
            self._chk(st_hist, (n_iter, b, 2, self.state_len), "st_hist")
        if rmse_hist is not None:
            self._chk(rmse_hist, (n_iter, b), "rmse_hist")
        rc = self.lib.hn_step(self.ctx, _ptr(wf), _ptr(res), _ptr(states), _ptr(k_sq), _ptr(src), src.shape[0], b,
                              int(n_iter), _ptr(res_hist), _ptr(wf_hist), _ptr(st_hist), _ptr(rmse_hist), self._stream())
        _lib.check(rc, self.ctx, "hn_step")

    # ---- training step (hn_train_grad / hn_adam_step) -------------------------------------------------
    def train_reserve(self, batch: int, n_unroll: int):
        _lib.check(self.lib.hn_train_reserve(self.ctx, int(batch), int(n_unroll)), self.ctx, "hn_train_reserve")

    def train_grad(self, weights, wf, res, states, k_sq, src, n_unroll: int, loss_scale: float = 1e4, input_grads: bool = False,
                   grad: Optional[torch.Tensor] = None) -> dict:
        """Loss and gradients of ``n_unroll`` unrolled solver iterations (hybridnet.py:399-409).  ``weights``: flat device blob
        (pack_weights order).  Returns loss [1], grad (same shape as weights), the wavefield / residual / state lists of
        ``n_steps(..., True, True)`` as stacked tensors, and with ``input_grads`` the gradients of the three inputs."""
        b, n_w = wf.shape[0], int(self.lib.hn_weight_count(8, self.depth, 2))
        self._chk(weights, (n_w,), "weights")
        self._chk(wf, (b, 2, self.n, self.n), "wavefield")
        self._chk(res, (b, 2, self.n, self.n), "residual")
        self._chk(states, (b, 2, self.state_len), "hidden state")
        self._chk(k_sq, (b, 1, self.n, self.n), "k_sq")
        self._chk(src, (src.shape[0], 2, self.n, self.n), "source")
        T = int(n_unroll)
        new = lambda *shape: torch.empty(shape, device=self.device, dtype=torch.float32)  # noqa: E731
        out = {"wavefields": new(T, b, 2, self.n, self.n), "residuals": new(T, b, 2, self.n, self.n), "states": new(T, b, 2, self.state_len),
               "loss": new(1), "grad": grad if grad is not None else new(n_w)}
        self._chk(out["grad"], (n_w,), "grad")
        g_in = [new(*wf.shape), new(*res.shape), new(*states.shape)] if input_grads else [None, None, None]
        rc = self.lib.hn_train_grad(self.ctx, _ptr(weights), _ptr(wf), _ptr(res), _ptr(states), _ptr(k_sq), _ptr(src), src.shape[0], b, T,
                                    float(loss_scale), _ptr(out["wavefields"]), _ptr(out["residuals"]), _ptr(out["states"]), _ptr(out["loss"]),
                                    _ptr(out["grad"]), _ptr(g_in[0]), _ptr(g_in[1]), _ptr(g_in[2]), self._stream())
        _lib.check(rc, self.ctx, "hn_train_grad")
        if input_grads:
            out.update(grad_wf=g_in[0], grad_res=g_in[1], grad_states=g_in[2])
        return out

    def adam_step(self, weights, grad, exp_avg, exp_avg_sq, step: int, lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                  weight_decay: float = 0.0, clip_value: float = 0.0, trainable: Optional[torch.Tensor] = None):
        """clip_grad_value_ + torch.optim.Adam step (hybridnet.py:172-176, 250-258) on caller-owned flat device tensors, in place."""
        n = weights.numel()
        for t, name in ((weights, "weights"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            self._chk(t, (n,), name)
        if trainable is not None and (trainable.dtype != torch.uint8 or trainable.numel() != n or trainable.device != self.device):
            raise ValueError("trainable must be a uint8 device tensor with one entry per weight")
        rc = self.lib.hn_adam_step(self.ctx, _ptr(weights), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(trainable), n, float(lr),
                                   float(betas[0]), float(betas[1]), float(eps), float(weight_decay), float(clip_value), int(step), self._stream())
        _lib.check(rc, self.ctx, "hn_adam_step")

    # ---- rows of a replay buffer (hn_rows_gather / hn_rows_scatter) ----------------------------------------------------
    def _rows_args(self, buffers, slots):
        n = len(buffers)
        cap = buffers[0].shape[0]
        for b in buffers:
            if b.dtype != torch.float32 or b.device != self.device or not b.is_contiguous() or b.shape[0] != cap:
                raise ValueError("replay-buffer fields must be contiguous fp32 [capacity, ...] tensors on the engine's device")
        slots = np.ascontiguousarray(np.asarray(slots).reshape(-1), dtype=np.int32)
        rows = (ctypes.c_int64 * n)(*[b[0].numel() for b in buffers])
        return n, cap, slots, rows, (ctypes.c_void_p * n)(*[b.data_ptr() for b in buffers])

    def rows_gather(self, buffers, slots):
        """ReplayBuffer.sample's stack (replaybuffer.py:37-47): [len(slots), ...] copies of rows ``slots`` (host integers) of every buffer, one launch."""
        n, cap, slots, rows, bufs = self._rows_args(buffers, slots)
        out = [torch.empty((slots.size,) + tuple(b.shape[1:]), dtype=torch.float32, device=self.device) for b in buffers]
        outs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in out])
        rc = self.lib.hn_rows_gather(self.ctx, n, bufs, rows, cap, slots.ctypes.data_as(ctypes.c_void_p), int(slots.size), outs, self._stream())
        _lib.check(rc, self.ctx, "hn_rows_gather")
        return out

    def rows_scatter(self, buffers, slots, new_rows):
        """ReplayBuffer.append for a batch of slots (replaybuffer.py:29-30): ``buffers[f][slots[j]] = new_rows[f][j]``; ``None`` writes zeros, a tensor
        with ONE row is written to every slot.  One launch."""
        n, cap, slots, rows, bufs = self._rows_args(buffers, slots)
        keep, ptrs, strides = [], [], []
        for b, v in zip(buffers, new_rows):
            if v is None:
                ptrs.append(None); strides.append(0)
                continue
            if v.dtype != torch.float32 or v.device != self.device or tuple(v.shape[1:]) != tuple(b.shape[1:]) or v.shape[0] not in (1, slots.size):
                raise ValueError(f"new rows of shape {tuple(v.shape)} do not fit {slots.size} slots of {tuple(b.shape[1:])}")
            v = v.contiguous()
            keep.append(v)
            ptrs.append(v.data_ptr()); strides.append(0 if (v.shape[0] == 1 and slots.size > 1) else b[0].numel())
        rc = self.lib.hn_rows_scatter(self.ctx, n, bufs, rows, cap, slots.ctypes.data_as(ctypes.c_void_p), int(slots.size),
                                      (ctypes.c_void_p * n)(*ptrs), (ctypes.c_int64 * n)(*strides), self._stream())
        _lib.check(rc, self.ctx, "hn_rows_scatter")

    def set_train_forward_event(self, event: Optional[torch.cuda.Event], sumsq_host: Optional[torch.Tensor] = None) -> None:
        """hn_train_set_forward_event: ``event`` is recorded behind the forward sweep of every later ``train_grad`` (None clears it); ``sumsq_host``
        (a pinned fp32 host tensor) receives the [n_unroll, batch] table of per-sample sums of res^2 before the event fires."""
        handle, table, cap = None, None, 0
        if event is not None:
            if not event.cuda_event:      # (torch creates the HIP event lazily: recording it once makes the handle exist)
                event.record(torch.cuda.current_stream(self.device))
            handle = ctypes.c_void_p(event.cuda_event)
        if sumsq_host is not None:
            if sumsq_host.dtype != torch.float32 or sumsq_host.device.type != "cpu" or not sumsq_host.is_pinned() or not sumsq_host.is_contiguous():
                raise ValueError("sumsq_host must be a contiguous pinned fp32 host tensor")
            table, cap = ctypes.c_void_p(sumsq_host.data_ptr()), sumsq_host.numel()
        _lib.check(self.lib.hn_train_set_forward_event(self.ctx, handle, table, cap), self.ctx, "hn_train_set_forward_event")
        self._fwd_event, self._fwd_sumsq = event, sumsq_host           # keep them alive while the library holds the handles

    PEEK = {"x": 0, "sig_mid": 1, "out": 2, "st_mid": 3, "u": 4, "dec_mid": 5, "y": 6, "inc_mid": 7, "g_x": 16, "g_out": 18, "g_u": 20, "g_y": 22}

    def train_peek(self, kind: str, level: int, batch: int) -> torch.Tensor:
        """Debug / test aid: one tape tensor or activation gradient of iteration 0 of the last train_grad call."""
        ch = 2 if kind == "st_mid" else 8
        m = self.n >> level
        out = torch.empty(batch, ch, m, m, device=self.device, dtype=torch.float32)
        got = self.lib.hn_train_peek(self.ctx, self.PEEK[kind], int(level), _ptr(out), out.numel(), self._stream())
        if got < 0:
            _lib.check(int(got), self.ctx, "hn_train_peek")
        return out

    # ---- measurement hooks ---------------------------------------------------------------
    KERNEL_IDS = 37

    @staticmethod
    def kernel_name(kid: int) -> str:
        if kid == 0:
            return "inc"
        if 1 <= kid <= 18:
            return ("conv_signal", "conv_state", "down")[(kid - 1) % 3] + str((kid - 1) // 3)
        if kid == 19:
            return "bottleneck"
        if 20 <= kid <= 31:
            return ("up", "decode")[(kid - 20) % 2] + str((kid - 20) // 2)
        return {32: "spectral_cols", 33: "spectral_rows", 34: "deep", 35: "spectral_pair", 36: "inc_conv_signal0"}[kid]

    def profile_enable(self, kernel_ids=None):
        """Bracket the selected kernels (None = all, [] = none) with HIP events on the launch stream."""
        mask = (1 << self.KERNEL_IDS) - 1 if kernel_ids is None else sum(1 << k for k in kernel_ids)
        _lib.check(self.lib.hn_profile_enable(self.ctx, ctypes.c_uint64(mask)), self.ctx, "hn_profile_enable")

    def profile_min(self) -> dict:
        """{kernel name: shortest bracketed launch in ms} for the interval closed by the last profile_collect()."""
        ms = (ctypes.c_double * self.KERNEL_IDS)()
        _lib.check(self.lib.hn_profile_min(self.ctx, ms, self.KERNEL_IDS), self.ctx, "hn_profile_min")
        return {self.kernel_name(i): ms[i] for i in range(self.KERNEL_IDS) if ms[i] > 0}

    def profile_stride(self, every_nth: int):
        """Bracket only every n-th launch of the selected kernels (an event pair costs microseconds of gap)."""
        _lib.check(self.lib.hn_profile_stride(self.ctx, int(every_nth)), self.ctx, "hn_profile_stride")

    def profile_collect(self) -> dict:
        """{kernel name: (total ms, launches)} since the last collect."""
        ms = (ctypes.c_double * self.KERNEL_IDS)()
        cnt = (ctypes.c_int64 * self.KERNEL_IDS)()
        _lib.check(self.lib.hn_profile_collect(self.ctx, ms, cnt, self.KERNEL_IDS), self.ctx, "hn_profile_collect")
        return {self.kernel_name(i): (ms[i], int(cnt[i])) for i in range(self.KERNEL_IDS) if cnt[i]}
