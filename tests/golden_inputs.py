"""Deterministic inputs shared by tests/golden/make_golden.py and the parity tests.

``numpy.random.default_rng`` (PCG64) streams are stable across numpy releases,
so the inputs of the teacher-forced golden cases are re-created here from
their seeds instead of being committed.
"""
import numpy as np

from helmnet_amd.phantoms import readme_sos, ring_sos_batch  # noqa: F401 (re-exported)


def state_len(n: int, depth: int = 4) -> int:
    return sum((n // 2 ** d) ** 2 for d in range(depth))


def teacher_inputs(n: int, b: int, seed: int):
    """White-noise wavefield / residual / hidden state + random SoS in [1, 2].

    White noise excites every spatial frequency, so it is the harshest input
    for the spectral operator; magnitudes are those the solver sees
    (|wf| ~ 1, 1e3*|res| ~ 5, |state| ~ 1).
    """
    rng = np.random.default_rng(seed)
    f32 = np.float32
    return {
        "wf": (0.5 * rng.standard_normal((b, 2, n, n))).astype(f32),
        "res": (5e-3 * rng.standard_normal((b, 2, n, n))).astype(f32),
        "states": (0.5 * rng.standard_normal((b, 2, state_len(n)))).astype(f32),
        "sos": (1.0 + rng.random((b, 1, n, n))).astype(f32),
    }


def long_inputs(tag: str):
    """Inputs of the long free-run fixtures (tests/golden/make_long_golden.py -> long_run.npz).

    cfg2: BASELINE configs[1] -- README map + 4 ring maps, 256^2, source [30, 128], 1000 iterations
    cfg4: BASELINE configs[3] -- one 512^2 ring map, source [450, 256], 2000 iterations
    cfg5: BASELINE configs[4] -- synthetic transcranial map (full 1.87x contrast) + arc source map, 512^2
    """
    from helmnet_amd.phantoms import arc_source_map, skull_sos
    if tag == "cfg2":
        return {"sos": np.concatenate([readme_sos(), ring_sos_batch(256, 4, seed=11)]), "loc": [30, 128]}
    if tag == "cfg4":
        return {"sos": ring_sos_batch(512, 1, seed=12), "loc": [450, 256]}
    if tag == "cfg5":
        return {"sos": skull_sos(512, 1, seed=0), "src_map": arc_source_map(512)}
    raise KeyError(tag)


def metric_inputs(seed: int = 2024):
    """Seeded inputs of the accuracy-metric fixtures (tests/golden/make_golden_r2.py): smooth complex fields [3, 2,
    96, 96] (re, im), a reference that differs by a few percent, a 0/1 mask and a two-frame stream."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(96, dtype=np.float32), np.arange(96, dtype=np.float32), indexing="ij")
    r = np.sqrt((yy - 82) ** 2 + (xx - 48) ** 2) + 1.0
    base = np.stack([np.cos(r) / np.sqrt(r), np.sin(r) / np.sqrt(r)]).astype(np.float32)     # outgoing wave from [82, 48]
    sample = np.stack([base * (1 + 0.1 * b) + 0.02 * rng.standard_normal(base.shape).astype(np.float32) for b in range(3)])
    reference = np.stack([base * (1.3 - 0.2 * b) + 0.02 * rng.standard_normal(base.shape).astype(np.float32) for b in range(3)])
    reference[:, 1] *= -1.0                                                                  # stored conjugated, as k-Wave's
    mask = np.zeros((96, 96), np.float32)
    mask[20:70, 15:80] = 1.0
    stream = np.stack([0.5 * sample, sample], 1).astype(np.float32)                          # [B, T, 2, H, W]
    return {"sample": sample.astype(np.float32), "reference": reference.astype(np.float32), "mask": mask, "stream": stream}
