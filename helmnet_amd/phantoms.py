"""Synthetic sound-speed (SoS) maps for tests and benchmarks (pure numpy, host side).

The paper's test set is a Google-Drive download that is not available offline,
so the benchmark inputs are regenerated from the *distribution* the reference
draws its training/test maps from: background 1.0 plus one closed harmonic
ring ("skull") of value 1.5..2.0 and thickness 2..9 px
(reference helmnet/dataloaders.py:115-156, ``EllipsesDataset._make_ellipsoid``;
that code rasterises with cv2.polylines, here the curve is densely sampled and
stamped with a disc of the same thickness -- cv2 is not needed).
"""
from __future__ import annotations

import numpy as np


def ring_sos(n: int, rng: np.random.Generator) -> np.ndarray:
    """One [n, n] float32 SoS map: background 1.0 + a random harmonic ring."""
    avg = np.array([1.0, 0.0, 0.0, 0.0])
    std = np.array([0.1, 0.05, 0.025, 0.01])
    a_x = avg + rng.standard_normal(4) * std
    a_y = avg + rng.standard_normal(4) * std
    ph_x = rng.standard_normal(4) * (np.pi / 16)
    ph_y = rng.standard_normal(4) * (np.pi / 16)
    t = np.linspace(0.0, 2 * np.pi, num=max(360, 8 * n), endpoint=True)
    x = sum(np.sin(t * (i + 1) + ph_x[i]) * a_x[i] for i in range(4))
    y = sum(np.cos(t * (i + 1) + ph_y[i]) * a_y[i] for i in range(4))
    x = (x + 2) / 4 * n  # column coordinate (cv2 point order is (x, y))
    y = (y + 2) / 4 * n  # row coordinate
    thickness = int(2 + rng.random() * 8)
    boost = 0.5 + rng.random() * 0.5
    img = np.zeros((n, n), np.float32)
    r = thickness / 2.0
    k = int(np.ceil(r))
    cx, cy = np.rint(x).astype(np.int64), np.rint(y).astype(np.int64)
    for dy in range(-k, k + 1):
        for dx in range(-k, k + 1):
            if dx * dx + dy * dy <= r * r + 0.25:
                yy, xx = cy + dy, cx + dx
                ok = (yy >= 0) & (yy < n) & (xx >= 0) & (xx < n)
                img[yy[ok], xx[ok]] = 1.0
    return (1.0 + img * boost).astype(np.float32)


def ring_sos_batch(n: int, batch: int, seed: int = 0) -> np.ndarray:
    """[batch, 1, n, n] float32 ring phantoms (BASELINE.json config 2 inputs)."""
    rng = np.random.default_rng(seed)
    return np.stack([ring_sos(n, rng) for _ in range(batch)])[:, None]


def readme_sos(n: int = 256) -> np.ndarray:
    """[1, 1, 256, 256] map of the reference README / test.py:17-18 example:
    ones with a rectangle whose speed ramps linearly from 2 down to 1."""
    assert n == 256
    sos = np.ones((n, n), np.float32)
    sos[100:170, 30:240] = np.tile(np.linspace(2, 1, 210), (70, 1)).astype(np.float32)
    return sos[None, None]


def smooth_random_sos(n: int, batch: int, seed: int = 1) -> np.ndarray:
    """[batch, 1, n, n]: 1 + U(0,1) blurred with an 8x8 box (values stay in [1, 2]);
    BASELINE.json config 3 inputs."""
    rng = np.random.default_rng(seed)
    u = rng.random((batch, n + 7, n + 7), dtype=np.float32)
    c = np.cumsum(np.cumsum(np.pad(u, ((0, 0), (1, 0), (1, 0))), axis=1), axis=2)
    box = (c[:, 8:, 8:] - c[:, :-8, 8:] - c[:, 8:, :-8] + c[:, :-8, :-8]) / 64.0
    return (1.0 + box[:, None, :n, :n]).astype(np.float32)


def skull_sos(n: int = 512, batch: int = 1, seed: int = 0, boost: float = 0.87, brain: float = 0.0) -> np.ndarray:
    """[batch, 1, n, n] synthetic transcranial phantoms for BASELINE.json configs[4] (the CQ500-derived maps of
    the paper are not redistributable): water 1.0 and an elliptical skull shell of relative sound speed
    1 + boost (0.87: 2800 / 1500 m/s) with smooth tables and a slower diploe.  ``brain`` > 0 adds soft-tissue
    heterogeneity inside (out of the network's training distribution, dataloaders.py:115-156, which is a
    homogeneous interior behind a constant-boost ring).  Values stay in [1, 2]; the PML margin stays water."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), indexing="ij")
    out = np.empty((batch, 1, n, n), np.float32)
    for b in range(batch):
        cy, cx = n * (0.52 + 0.03 * rng.standard_normal()), n * (0.5 + 0.03 * rng.standard_normal())
        ay, ax = n * (0.30 + 0.02 * rng.random()), n * (0.24 + 0.02 * rng.random())
        th = rng.uniform(-0.2, 0.2)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        ang = np.arctan2(v, u)
        wobble = 1.0 + 0.03 * np.sin(3 * ang + rng.uniform(0, 6.28)) + 0.02 * np.sin(5 * ang + rng.uniform(0, 6.28))
        r = np.sqrt((u / ax) ** 2 + (v / ay) ** 2) / wobble          # 1 on the outer table
        thick = 0.10 + 0.03 * np.sin(2 * ang + rng.uniform(0, 6.28))  # shell thickness in units of r
        edge = 0.012
        outer = 1.0 / (1.0 + np.exp((r - 1.0) / edge))
        inner = 1.0 / (1.0 + np.exp((r - (1.0 - thick)) / edge))
        shell = outer - inner                                         # 1 inside the bone
        mid = np.exp(-(((r - (1.0 - 0.5 * thick)) / (0.25 * thick)) ** 2))
        bone = boost * (1.0 - 0.29 * mid)                             # cortical tables faster than the diploe
        soft = brain * (0.5 + 0.5 * np.sin(6.28 * (2.0 * u / n + 0.3)) * np.sin(6.28 * (1.5 * v / n + 0.1)))
        out[b, 0] = 1.0 + shell * bone + inner * soft
    return np.clip(out, 1.0, 2.0).astype(np.float32)


def arc_source_mask(n: int, arc_pos, radius: float, diameter: float, focus_pos) -> np.ndarray:
    """[n, n] 0/1 mask of a focused bowl transducer in 2-D: the pixels on the circle of ``radius`` through
    ``arc_pos`` (its mid point, [row, col]) curved towards ``focus_pos``, within the aperture ``diameter``
    (chord length).  Geometry of matlab/skull_example.m:80 (k-Wave ``makeArc``, not part of the reference
    checkout); the rasterisation rule here is: |distance to the centre of curvature - radius| < 0.5."""
    ap, fp = np.asarray(arc_pos, np.float64), np.asarray(focus_pos, np.float64)
    axis = (fp - ap) / np.linalg.norm(fp - ap)
    centre = ap + radius * axis
    half = np.arcsin(min(1.0, diameter / (2.0 * radius)))
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64), indexing="ij")
    dy, dx = yy - centre[0], xx - centre[1]
    r = np.hypot(dy, dx)
    cosang = (dy * (-axis[0]) + dx * (-axis[1])) / np.maximum(r, 1e-9)
    return ((np.abs(r - radius) < 0.5) & (cosang >= np.cos(half))).astype(np.float32)


def arc_source_map(n: int = 512, arc_pos=(430, 380), radius: float = 122.0, diameter: float = 122.0,
                   focus_pos=(320, 256), amplitude: float = 10.0) -> np.ndarray:
    """[1, 2, n, n] source map of the transcranial example (BASELINE.json configs[4]): amplitude x arc mask in
    BOTH channels -- support_functions.py:321-326 hands ``10 * src`` of shape [1, n, n] to ``set_domain_size``,
    which broadcasts over the real and the imaginary channel in ``get_residual`` (hybridnet.py:556)."""
    m = amplitude * arc_source_mask(n, [p * n / 512.0 for p in arc_pos], radius * n / 512.0, diameter * n / 512.0,
                                    [p * n / 512.0 for p in focus_pos])
    return np.stack([m, m])[None].astype(np.float32)
