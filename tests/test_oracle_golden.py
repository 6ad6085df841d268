"""Pin the CPU oracle (oracle/helmnet_oracle.py) against the golden vectors that
tests/golden/make_golden.py produced by running the reference itself.
CPU only; runs in the default ``-m "not gpu"`` suite."""
import numpy as np
import pytest
import torch

from golden_inputs import teacher_inputs
from helmnet_amd.phantoms import readme_sos, ring_sos_batch
from oracle import helmnet_oracle as O

SRC = {96: [82, 48], 256: [30, 128], 512: [450, 256]}


def tables(n):
    return O.SpectralTables(n, 8, 2, 1.0)


@pytest.mark.parametrize("n", [96, 256, 512])
def test_setup_tables_bit_exact(n, g_setup):
    t = tables(n)
    g = g_setup
    assert np.array_equal(t.kx[0, 0, :, 1].numpy(), g[f"n{n}_kx_row"])
    assert np.array_equal(t.ky[0, :, 0, 1].numpy(), g[f"n{n}_ky_col"])
    assert np.array_equal(t.kx_sq[0, 0, :, 0].numpy(), g[f"n{n}_kxsq_row"])
    assert np.array_equal(t.ky_sq[0, :, 0, 0].numpy(), g[f"n{n}_kysq_col"])
    for nm in ("ax", "bx"):
        assert np.array_equal(getattr(t, nm)[0, 0].numpy(), g[f"n{n}_{nm}_row"])
    for nm in ("ay", "by"):
        assert np.array_equal(getattr(t, nm)[0, :, 0].numpy(), g[f"n{n}_{nm}_col"])
    assert np.array_equal(t.sigmas[0, 0].numpy(), g[f"n{n}_sigma_x_row"])
    assert np.array_equal(t.sigmas[1, :, 0].numpy(), g[f"n{n}_sigma_y_col"])
    # SURVEY A.1 hand-checkable values
    assert t.kx[0, 0, n // 2, 1].item() == pytest.approx(-np.pi, rel=1e-7)
    assert t.bx[0, 0, 0].numpy() == pytest.approx([-0.12, -0.16], abs=1e-7)
    assert t.ax[0, 0, 0].numpy() == pytest.approx([-0.008, -0.044], abs=1e-7)
    src = O.point_source_map(n, SRC[n], 10.0)
    assert np.allclose(src[0, :, SRC[n][0], SRC[n][1]].numpy(), g[f"n{n}_source_peak"], atol=1e-5)
    if n == 96:
        assert np.allclose(src.numpy(), g["n96_source"], atol=2e-6)
        assert np.array_equal(t.ax.numpy(), g["n96_ax_full"])
        assert np.array_equal(t.by.numpy(), g["n96_by_full"])


def _cmp(name, got, g, n, tol):
    got = got.contiguous().numpy()
    scale = float(g[f"n{n}_{name}_absmax"])
    if n == 96:
        err = np.abs(got - g[f"n{n}_{name}"]).max()
    elif got.ndim == 4:
        err = max(np.abs(got[:, :, :40, :40] - g[f"n{n}_{name}_crop"]).max(),
                  np.abs(got[:, :, 3::7, 5::11] - g[f"n{n}_{name}_stride"]).max())
    else:
        err = np.abs(got[:, :, 1::37] - g[f"n{n}_{name}_stride"]).max()
    assert err <= tol * scale, (name, n, err, scale)


@pytest.mark.parametrize("n,b", [(96, 2), (256, 2), (512, 1)])
def test_teacher_forced_ops(n, b, g_teacher, weights):
    """One get_residual / HybridNet.forward / single_step on identical inputs.
    fp32 tolerance 1e-5 * max|.|; the oracle uses the
    same ATen kernels as the reference so it is in fact much closer."""
    ti = {k: torch.from_numpy(v) for k, v in teacher_inputs(n, b, seed=1000 + n).items()}
    t = tables(n)
    src = O.point_source_map(n, SRC[n], 10.0)
    k_sq, _ = O.get_initials(ti["sos"], 1.0)
    _cmp("lap", O.apply_laplacian(ti["wf"], t), g_teacher, n, 1e-5)
    _cmp("residual", O.get_residual(ti["wf"], k_sq, src, t), g_teacher, n, 1e-5)
    st = O.unflatten_states(ti["states"], n, 4)
    sig = t.sigmas.unsqueeze(0).repeat(b, 1, 1, 1)
    d, st_new = O.unet_forward(torch.cat([ti["wf"], 1e3 * ti["res"], sig], 1), st, weights)
    _cmp("unet_d", d, g_teacher, n, 1e-5)
    _cmp("states_new", O.flatten_states(st_new), g_teacher, n, 1e-5)
    wf2, res2, _ = O.single_step(ti["wf"], k_sq, ti["res"], st, weights, src, t)
    _cmp("step_wf", wf2, g_teacher, n, 1e-5)
    _cmp("step_res", res2, g_teacher, n, 1e-5)


def test_free_run_cfg1_homogeneous(g_free, weights):
    """BASELINE.json config 1: 256^2, sos == 1, source [30,128], 100 iterations."""
    t = tables(256)
    src = O.point_source_map(256, [30, 128], 10.0)
    out = O.solve(torch.ones(1, 1, 256, 256), weights, src, t, 100)
    trace = torch.stack(out["trace"]).numpy()
    assert np.allclose(trace, g_free["cfg1_rmse"], rtol=1e-2)
    assert trace[0, 0] == pytest.approx(6.154e-3, rel=2e-3)     # BASELINE.md section 2 known answers
    assert trace[-1, 0] == pytest.approx(3.018e-5, rel=2e-2)
    assert np.abs(out["wavefield"].numpy() - g_free["cfg1_wf_it100"]).max() <= 1e-4
    assert out["wavefield"].abs().max().item() == pytest.approx(2.4794, abs=2e-3)


def test_free_run_ring96(g_free, weights):
    """Native 96^2 domain, batch of 3 ring phantoms, 200 iterations."""
    t = tables(96)
    src = O.point_source_map(96, [82, 48], 10.0)
    sos = torch.from_numpy(ring_sos_batch(96, 3, seed=7))
    out = O.solve(sos, weights, src, t, 200)
    assert np.allclose(torch.stack(out["trace"]).numpy(), g_free["ring96_rmse"], rtol=2e-2)
    assert np.abs(out["wavefield"].numpy() - g_free["ring96_wf_it200"]).max() <= 1e-4


def test_free_run_line_source(g_free, weights):
    """examples/simple_scattering.py problem: rectangle 1.5, line source map."""
    t = tables(256)
    sos = np.ones((256, 256), np.float32)
    sos[100:170, 30:240] = 1.5
    smap = np.zeros((1, 2, 256, 256), np.float32)
    smap[0, 0, 30, 120:130] = 1
    out = O.solve(torch.from_numpy(sos)[None, None], weights, torch.from_numpy(smap), t, 100)
    assert np.allclose(torch.stack(out["trace"]).numpy(), g_free["scatter_rmse"], rtol=1e-2)
    assert np.abs(out["wavefield"].numpy() - g_free["scatter_wf_it100"]).max() <= 1e-4


def test_readme_known_answers(g_free):
    """SURVEY section 4 table for the README problem (values measured on the reference)."""
    tr = g_free["readme_rmse"]
    assert tr[0, 0] == pytest.approx(6.1573e-3, rel=1e-3)
    assert tr[99, 0] == pytest.approx(1.4162e-4, rel=2e-2)
    assert readme_sos().shape == (1, 1, 256, 256)


def test_plane_wave_kat():
    """Analytic known answer needing no golden: in the non-PML interior the
    Laplacian of exp(i*2*pi*m*x/N) is -(2*pi*m/N)^2 times itself."""
    n, m = 256, 5
    x = torch.arange(n, dtype=torch.float64)
    ph = 2 * np.pi * m * x / n
    u = torch.stack([torch.cos(ph), torch.sin(ph)], 0).unsqueeze(1).repeat(1, n, 1).unsqueeze(0).float()
    lap = O.apply_laplacian(u, tables(n))
    want = -(2 * np.pi * m / n) ** 2 * u
    assert (lap - want)[:, :, 8:-8, 8:-8].abs().max().item() < 5e-6
