"""SURVEY 8(f) "next" rows: dataset container, accuracy metrics (CPU) and GMRES on the HIP operator (GPU)."""
import numpy as np
import pytest
import torch


def test_dataset_roundtrip(tmp_path):
    from helmnet_amd.datasets import EllipsesDataset, get_dataset
    ds = EllipsesDataset()
    assert ds.all_sos == [] and len(ds) == 0
    ds.make_dataset(num_ellipses=5, imsize=64, seed=3)
    ds.sos_maps_to_tensor()
    assert len(ds) == 5 and ds[2].shape == (1, 64, 64) and ds[2].dtype == torch.float32
    assert float(ds.all_sos.min()) == 1.0 and 1.5 <= float(ds.all_sos.max()) <= 2.0
    ds.save_dataset(str(tmp_path / "maps.npy"))
    again = get_dataset(str(tmp_path / "maps.npy"))
    assert torch.equal(again.all_sos, ds.all_sos)
    torch.save(ds, tmp_path / "set.ph")
    ph = get_dataset(str(tmp_path / "set.ph"), source_location="cuda:7", destination="cpu")
    assert torch.equal(ph[4], ds[4])


def test_metrics_match_reference_formulas():
    from helmnet_amd.metrics import as_complex, difference_to_reference, last_frame_difference, normalize_wavefield
    g = torch.Generator().manual_seed(0)
    wf = torch.randn(3, 2, 96, 96, generator=g)
    c = as_complex(wf)
    n = normalize_wavefield(c, [82, 48])
    assert torch.allclose(n[:, 82, 48], torch.ones(3, dtype=n.dtype))
    assert torch.allclose(normalize_wavefield(c[0], [82, 48])[82, 48], torch.ones((), dtype=n.dtype))
    # identical fields (reference stored conjugated, as k-Wave's convention) -> zero difference
    diff, s, r = difference_to_reference(c, torch.conj(c))
    assert diff.shape == (3, 76, 76) and float(diff.max()) < 1e-6
    # a global complex scale is removed by the source normalisation
    diff2, _, _ = difference_to_reference(c * (0.3 - 2j), torch.conj(c))
    assert float(diff2.max()) < 1e-5
    stream = torch.stack([wf * 0.5, wf], 1)               # [B, T, 2, H, W]
    l_inf, rmse = last_frame_difference(stream, torch.conj(c) * 1.7)
    assert l_inf.shape == (3,) and float(l_inf.max()) < 1e-5 and float(rmse.max()) < 1e-5
    mask = torch.zeros(96, 96); mask[20:70, 20:70] = 1
    diffm, _, _ = difference_to_reference(c + 0.01, torch.conj(c), mask=mask)
    assert float(diffm.max()) <= 1.0


@pytest.mark.gpu
def test_gmres_on_hip_operator_agrees_with_learned_solver():
    """Both solvers drive the SAME residual to zero, so their wavefields must agree to the level of
    their residuals; GMRES' reported norm must equal the true residual RMSE."""
    from helmnet_amd import IterativeSolver
    from helmnet_amd.gmres import gmres
    from helmnet_amd.phantoms import ring_sos_batch
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to("cuda:0")
    s.set_domain_size(96, source_location=[82, 48])
    sos = torch.from_numpy(ring_sos_batch(96, 2, seed=11)).to("cuda:0")
    out = gmres(s, sos, restart=30, max_outer=40, tol=2e-4)
    k_sq, _ = s.get_initials(sos)
    true_rmse = s.engine().rmse(s.get_residual(out["wavefield"], k_sq.contiguous()))
    assert torch.allclose(true_rmse, out["residual_norms"][-1].to(true_rmse.device), rtol=5e-2, atol=2e-5)
    assert float(true_rmse.max()) < 1e-3 and out["iterations"] > 10
    first = out["residual_norms"][0]
    assert float((out["residual_norms"][-1] / first).max()) < 1e-1          # it converges
    learned = s.forward(sos, num_iterations=400, residuals="norms")
    assert float(learned["residual_norms"][-1].max()) < 2e-4
    a, b = out["wavefield"], learned["wavefields"][0]
    rel = (a - b).abs().amax(dim=(1, 2, 3)) / b.abs().amax(dim=(1, 2, 3))
    assert float(rel.max()) < 0.05, rel


@pytest.mark.gpu
def test_convergence_to_tolerance_on_a_transcranial_phantom():
    """BASELINE.json configs[4] shape (512^2, set_domain_size, run until the worst residual RMSE is below a
    tolerance): the chunked loop stops at the first check below tol and equals one uninterrupted forward()."""
    import torch
    from helmnet_amd import IterativeSolver
    from helmnet_amd.phantoms import skull_sos
    dev = torch.device("cuda:0")
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to(dev)
    s.set_domain_size(512, source_location=[51, 256])
    sos_np = skull_sos(512, 2, seed=0, boost=0.4)
    assert sos_np.min() == 1.0 and 1.3 < sos_np.max() <= 2.0 and (sos_np[:, :, :40] == 1.0).all()   # PML margin stays water
    sos = torch.from_numpy(sos_np).to(dev)
    out = s.solve_to_tolerance(sos, tol=1e-4, max_iterations=1000, check_every=50)
    assert out["converged"] and out["iterations"] <= 400 and out["iterations"] % 50 == 0
    rm = out["residual_norms"]
    assert rm.shape == (out["iterations"], 2) and float(rm[-1].max()) < 1e-4 <= float(rm[-51].max())
    ref = s.forward(sos, num_iterations=out["iterations"], residuals="norms")
    assert torch.equal(ref["wavefields"][0], out["wavefield"])                       # same arithmetic, chunked or not
    assert torch.allclose(ref["residual_norms"], rm, rtol=1e-5, atol=0)              # per-sample sums are float atomics
    slow = s.solve_to_tolerance(sos, tol=1e-12, max_iterations=100, check_every=30)
    assert not slow["converged"] and slow["iterations"] == 100 and slow["residual_norms"].shape[0] == 100
