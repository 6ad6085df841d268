"""Host-side mirror of the reference's stateful UNet (``HybridNet``).

Mirrors reference helmnet/architectures.py: ``OutConv`` (:47-60), ``DoubleConv`` (:63-84),
``EncoderBlock`` (:186-252) and ``HybridNet`` (:317-465) -- same constructor arguments, same
sub-module / parameter names (so the shipped checkpoint's ``f.*`` state_dict loads unchanged),
same state bookkeeping API (``clear_states / get_states / set_states / flatten_state /
unflatten_state / init_by_size``, ``enc[d].state``, ``enc[d].domain_size``).

The torch ``nn.Conv2d`` / ``nn.PReLU`` / ``nn.ConvTranspose2d`` objects below are PARAMETER CONTAINERS: they are never
called.  ``HybridNet.forward`` hands the packed weights and the flat hidden state to libhelmnet_hip.so (``hn_unet``),
which runs the whole network as fused HIP kernels; the per-iteration solver loop bypasses even that and uses
``hn_step``.  ``DoubleConv`` / ``OutConv`` / ``EncoderBlock`` called on their own go through the library's standalone
entry points (``hn_double_conv`` / ``hn_out_conv`` / ``hn_conv8x8``) for the channel shapes the UNet is made of.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .engine import Engine, module_engine, pack_weights

_IMPLEMENTED = ("prelu", "relu", "leakyrelu", "celu", "tanh", "gelu", "tanhshrink", "softplus")


def getActivationFunction(act_function_name: str, features=None, end=False) -> nn.Module:
    """architectures.py:5-44: every name the reference knows except 'relu_batchnorm' (BatchNorm2d parameters and
    running statistics are not part of the kernels' weight blob); unknown names raise NotImplementedError as in
    the reference.  The returned modules are parameter containers / markers only (see module docstring)."""
    name = act_function_name.lower()
    if name == "prelu":
        return nn.PReLU()
    if name == "relu":
        return nn.ReLU(inplace=True)
    if name == "leakyrelu":
        return nn.LeakyReLU(inplace=True)
    if name == "celu":
        return nn.CELU(inplace=True)
    if name == "tanh":
        return nn.Tanh()
    if name == "gelu":
        return nn.GELU()
    if name == "tanhshrink":
        return nn.Tanhshrink()
    if name == "softplus":
        return nn.Softplus()
    raise NotImplementedError("Unknown activation function {} (implemented: {})".format(act_function_name, _IMPLEMENTED))


_ACT_CONST_SLOPE = {"relu": 0.0, "leakyrelu": 0.01}


class OutConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        """architectures.py:57-60, through hn_out_conv (8 -> 2 channels, the shape the UNet uses)."""
        if tuple(self.conv.weight.shape[:2]) != (2, 8):
            raise NotImplementedError("OutConv runs on the HIP library for 8 -> 2 channels only")
        return module_engine(x.device).out_conv(x, self.conv.weight, self.conv.bias)


class DoubleConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, mid_channels=None, activation_fun="relu"):
        super().__init__()
        mid_channels = out_channels if mid_channels is None else mid_channels
        self.activation_fun = activation_fun
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1),
            getActivationFunction(activation_fun, mid_channels),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1),
        )

    def forward(self, x):
        """architectures.py:83-84, through hn_double_conv (channel shapes of the UNet: 6/8/10/16 -> 8 -> 8 and 10 -> 2 -> 2)."""
        c1, act, c2 = self.double_conv[0], self.double_conv[1], self.double_conv[2]
        if c1.out_channels != c2.out_channels:
            raise NotImplementedError("DoubleConv runs on the HIP library with mid_channels == out_channels only")
        name = self.activation_fun.lower()
        slope = act.weight if isinstance(act, nn.PReLU) else _ACT_CONST_SLOPE.get(name, 0.0)
        return module_engine(x.device).double_conv(x, c1.weight, c1.bias, slope, c2.weight, c2.bias, name)


class EncoderBlock(nn.Module):
    def __init__(self, num_features: int, state_size=2, activation_function="prelu", use_state=True, domain_size=0):
        super().__init__()
        self.state_size, self.use_state, self.domain_size, self.num_features = state_size, use_state, domain_size, num_features
        # architectures.py:202-218: without state conv_signal sees the features only and there is no conv_state
        self.conv_signal = DoubleConv(num_features + state_size * use_state, num_features, activation_fun=activation_function)
        self.down = nn.Conv2d(num_features, num_features, kernel_size=8, padding=3, stride=2)
        if use_state:
            self.conv_state = DoubleConv(num_features + state_size, state_size, activation_fun=activation_function)
        self.state: Optional[torch.Tensor] = None

    def set_state(self, state):
        self.state = state

    def get_state(self):
        return self.state

    def clear_state(self, x):
        self.state = torch.zeros([x.shape[0], 2, self.domain_size, self.domain_size], device=x.device)

    def forward(self, x):
        """architectures.py:240-252 on the library's standalone entry points (hn_double_conv, hn_conv8x8)."""
        if self.use_state:
            if self.state is None:
                raise ValueError("You must set or clear the state before using this module")
            output = self.conv_signal(torch.cat([x, self.state], 1))
            self.state = self.conv_state(torch.cat([output, self.state], 1))
        else:
            output = self.conv_signal(x)
        return output, module_engine(x.device).conv8x8(output, self.down.weight, self.down.bias, transposed=False)


class HybridNet(nn.Module):
    def __init__(self, activation_function: str, depth: int, domain_size: int, features: int, inchannels: int,
                 state_channels: int, state_depth: int):
        super().__init__()
        if features != 8 or state_channels != 2 or inchannels != 6:
            raise NotImplementedError("HIP kernels are built for features=8, state_channels=2, inchannels=6")
        if not 0 <= state_depth <= depth:
            raise ValueError(f"state_depth {state_depth} outside [0, depth = {depth}]")
        self.activation_function, self.depth, self.domain_size = activation_function, depth, domain_size
        self.features, self.inchannels = features, inchannels
        self.state_channels, self.state_depth = state_channels, state_depth
        self.init_by_size()
        self.inc = DoubleConv(inchannels, features, activation_fun=activation_function)
        self.enc = nn.ModuleList([
            EncoderBlock(features, state_size=state_channels, activation_function=activation_function,
                         use_state=d < state_depth, domain_size=self.states_dimension[d]) for d in range(depth)])
        self.decode = nn.ModuleList([
            DoubleConv(features + features * (i < depth), features, activation_fun=activation_function)
            for i in range(depth + 1)])
        self.up = nn.ModuleList([
            nn.ConvTranspose2d(features, features, kernel_size=8, padding=3, output_padding=0, stride=2)
            for _ in range(depth)])
        self.outc = OutConv(features, 2)
        self._engine: Optional[Engine] = None
        self._owns_engine = True
        self._unet_precision = None

    # ---- state bookkeeping (architectures.py:390-437) -----------------------------------
    def init_by_size(self):
        self.states_dimension = [self.domain_size // 2 ** x for x in range(self.depth)]
        self.total_state_length = sum(x ** 2 for x in self.states_dimension)
        self.state_boundaries, o = [], 0
        for s in self.states_dimension:
            self.state_boundaries.append([o, o + s * s])
            o += s * s

    def get_states(self, flatten=False):
        h = [enc.get_state() for enc in self.enc]
        return self.flatten_state(h) if flatten else h

    def clear_states(self, x):
        for enc in self.enc:
            enc.clear_state(x)

    def set_states(self, states, flatten=False):
        h = self.unflatten_state(states) if flatten else states
        for enc, state in zip(self.enc[: len(h)], h):
            enc.set_state(state)

    def adopt_states(self, new_flat):
        """Store the flat hidden state the library returned.  Levels without state (d >= state_depth) keep what they had:
        the reference never touches them (EncoderBlock.forward, architectures.py:250-251)."""
        h = self.unflatten_state(new_flat)
        for d, enc in enumerate(self.enc):
            if enc.use_state:
                enc.set_state(h[d])

    def to_engine_states(self, flat):
        """Levels without state (d >= state_depth) run in the kernels as zero-weight stateful levels (engine.pack_weights): their
        slot is multiplied by 0 and rewritten with 0.  Hand the library zeros there (0 * NaN from a diverged earlier run would
        otherwise reach conv_signal) and remember what the slot held; ``from_engine_states`` puts it back, so the slot reads as
        the reference leaves it: untouched (architectures.py:250-251).  ``flat`` is modified in place; returns the kept slices."""
        keep = {}
        for d, (a, b) in enumerate(self.state_boundaries):
            if d >= self.state_depth:
                keep[(a, b)] = flat[:, :, a:b].clone()
                flat[:, :, a:b] = 0
        return keep

    @staticmethod
    def from_engine_states(keep, *tensors):
        """Restore the stateless slots in flat state tensors [..., B, 2, L] (the state itself, a history of states)."""
        for t in tensors:
            if t is None:
                continue
            for (a, b), v in keep.items():
                t[..., a:b] = v

    def flatten_state(self, h_list):
        return torch.cat([x.reshape(x.shape[0], x.shape[1], -1) for x in h_list], 2)

    def unflatten_state(self, h_flatten):
        h, shp = [], h_flatten.shape
        for (a, b), size in zip(self.state_boundaries, self.states_dimension):
            h.append(h_flatten[:, :, a:b].reshape(shp[0], shp[1], size, size))
        return h

    # ---- engine plumbing -----------------------------------------------------------------
    def bind(self, engine: Engine):
        self._engine, self._owns_engine = engine, False

    def set_unet_precision(self, mode: str):
        """Standalone use (no IterativeSolver around it): arithmetic of the convolutions, see Engine.set_unet_precision."""
        self._unet_precision = mode
        if self._engine is not None and self._owns_engine:
            self._engine.set_unet_precision(mode)

    def weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def sync_weights(self, engine: Engine):
        """(Re-)upload the parameters if they changed since the last upload."""
        key = self.weights_key()
        if engine.weights_key != key:
            sd = {k: v for k, v in self.state_dict().items()}
            engine.load_weights(pack_weights(sd, self.depth, self.activation_function, self.state_depth), self.features, self.depth,
                                self.state_channels, self.activation_function)
            engine.weights_key = key

    def _get_engine(self, device) -> Engine:
        if self._engine is None or self._engine.device != torch.device(device):
            if not self._owns_engine and self._engine is not None:
                raise RuntimeError(f"input on {device} but the solver's engine lives on {self._engine.device}")
            self._engine = Engine(device)
        if self._owns_engine and self._engine.n != self.domain_size:
            # standalone use: the spectral tables are not needed, but the library wants a domain
            self._engine.set_domain(self.domain_size, 1, 0.0, 1.0)
        if self._owns_engine and self._unet_precision is not None and self._engine.unet_precision != self._unet_precision:
            self._engine.set_unet_precision(self._unet_precision)
        self.sync_weights(self._engine)
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, 6, N, N] -> d [B, 2, N, N]; updates every enc[d].state (architectures.py:439-465)."""
        if any(enc.state is None for enc in self.enc):
            raise ValueError("You must set or clear the state before using this module")
        eng = self._get_engine(x.device)
        flat = self.get_states(flatten=True).contiguous()
        keep = self.to_engine_states(flat)
        d, new_flat = eng.unet(x.contiguous(), flat)
        self.from_engine_states(keep, new_flat)
        self.adopt_states(new_flat)
        return d
