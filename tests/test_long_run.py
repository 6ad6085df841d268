"""Long free-run parity tier: the full iteration counts and batch
sizes of BASELINE.json configs[1] (256^2, B = 32, 1000 it), configs[3] (512^2, B = 16, 2000 it) and configs[4]
(512^2 transcranial map, arc source map, fp16 UNet / fp32 residual, convergence to tolerance) against the REFERENCE's
own float64 and float32 trajectories (tests/golden/long_run.npz, made by tests/golden/make_long_golden.py).

Bars.  The iteration amplifies rounding differences (the reference's own fp32 run drifts from its float64 run by up to
9e-4 after 1000 iterations, recorded per sample and checkpoint in the fixture), so an fp32 implementation is held to
    L_inf(wavefield - float64 trajectory) <= max(1e-4, 2 x the reference-fp32 deviation at that checkpoint)
on the stored probe grid -- the deviation being the LARGEST over the fixture's samples at that checkpoint: how far one
particular sample's fp32 rounding pattern happens to be amplified is a random quantity (the reference's own per-sample
deviations spread over a factor of 8 at iteration 300), its scale is what the reference run pins -- the residual RMSE trace to 2 % over the first 300 iterations (before the trajectories
decorrelate) and the converged RMSE floor to +-10 %.
"""
import os

import numpy as np
import pytest
import torch

from golden_inputs import long_inputs
from helmnet_amd.phantoms import ring_sos_batch
from oracle import helmnet_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def g_long():
    with np.load(os.path.join(REPO, "tests", "golden", "long_run.npz")) as z:
        return {k: z[k] for k in z.files}


def test_oracle_follows_the_float64_trajectory_first_100_iterations(g_long, weights):
    """The CPU restatement on one of the five config-2 maps (a ring phantom): 100 iterations, against the float64 reference probes."""
    li = long_inputs("cfg2")
    sos = torch.from_numpy(li["sos"][1:2])
    out = O.solve(sos, weights, O.point_source_map(256, li["loc"], 10.0), O.SpectralTables(256, 8, 2, 1.0), 100)
    st = int(g_long["cfg2_stride"])
    err = np.abs(out["wavefield"].numpy()[:, :, ::st, ::st] - g_long["cfg2_wf_it100"][1:2]).reshape(1, -1).max(1)
    bar = np.maximum(1e-4, 2 * g_long["cfg2_f32dev_probe_it100"].max())
    assert (err <= bar).all(), (err, bar)
    trace = torch.stack(out["trace"]).numpy()
    assert np.abs(trace / g_long["cfg2_rmse_f64"][:100, 1:2] - 1).max() <= 2e-2


def test_oracle_in_float64_is_the_float64_reference_trajectory(g_long, weights):
    """The leg the config-3 bar of the GPU tests stands on: ``O.solve`` evaluated in float64 IS the reference's float64 run --
    100 iterations of one config-2 map against the reference's float64 probes.  The fixture stores those probes rounded to float32, so agreement
    to float32 resolution (measured 3.2e-8 absolute = 1.7e-8 of the largest value; the bar is 1e-7 of it) is all that can be asked -- an fp32
    evaluation sits 1e-5 .. 1e-4 away (the test above), and a float64 run that merely starts from the fp32-built source map 1.2e-6."""
    li = long_inputs("cfg2")
    sos = torch.from_numpy(li["sos"][1:2]).double()
    w64 = {k: v.double() for k, v in weights.items()}
    torch.set_default_dtype(torch.float64)   # the reference's float64 run builds its source map under this default (make_long_golden.py: run)
    try:
        src = O.point_source_map(256, li["loc"], 10.0)
    finally:
        torch.set_default_dtype(torch.float32)
    out = O.solve(sos, w64, src, O.SpectralTables(256, 8, 2, 1.0, dtype=torch.float64), 100)
    st = int(g_long["cfg2_stride"])
    gold = g_long["cfg2_wf_it100"][1:2].astype(np.float64)
    got = out["wavefield"].numpy()[:, :, ::st, ::st]
    err, scale = np.abs(got - gold).max(), np.abs(gold).max()
    assert err <= 1e-7 * scale, (err, scale)
    trace = torch.stack(out["trace"]).numpy()
    assert np.abs(trace / g_long["cfg2_rmse_f64"][:100, 1:2] - 1).max() <= 1e-6


def _solver():
    from helmnet_amd import IterativeSolver
    s = IterativeSolver.from_exported_weights()
    s.freeze()
    s.to(DEV)
    return s


def _run_with_checkpoints(s, sos, checkpoints):
    """forward() to the first checkpoint, n_steps between the others (states carry over in s.f as in the reference)."""
    traces, fields, done = [], {}, 0
    k_sq = None
    for cp in checkpoints:
        if done == 0:
            o = s.forward(sos, num_iterations=cp, residuals="norms")
            k_sq = s.get_initials(sos)[0].contiguous()
        else:
            o = s.n_steps(wf, k_sq, res, cp - done, residuals="norms")
        wf, res = o["wavefields"][0], o["last_residual"]
        traces.append(o["residual_norms"].cpu().numpy())
        fields[cp] = wf.cpu().numpy()
        done = cp
    return np.concatenate(traces), fields


def _check(tag, g, trace, fields, n_gold, early=300, floor=slice(-50, None)):
    st = int(g[f"{tag}_stride"])
    r64 = g[f"{tag}_rmse_f64"]
    assert trace.shape[0] == r64.shape[0]
    assert np.isfinite(trace).all()
    rel = np.abs(trace[:early, :n_gold] / r64[:early] - 1).max()
    assert rel <= 2e-2, rel
    floor_got, floor_ref = np.median(trace[floor, :n_gold], 0), np.median(r64[floor], 0)
    assert (np.abs(floor_got / floor_ref - 1) <= 0.10).all(), (floor_got, floor_ref)
    report = {}
    for cp, wf in fields.items():
        err = np.abs(wf[:n_gold, :, ::st, ::st] - g[f"{tag}_wf_it{cp}"]).reshape(n_gold, -1).max(1)
        dev = g[f"{tag}_f32dev_probe_it{cp}"]      # what the REFERENCE's own fp32 run deviates from its float64 run, per sample
        # every sample within twice the reference's worst sample AND (r4) within four times what the reference's fp32 run deviates
        # on THAT sample (observed: 0.1 - 1.1 x, one sample of cfg2 at iteration 300 3.4 x: two fp32 evaluations of a chaotic iteration)
        bar = np.minimum(np.maximum(1e-4, 2 * dev.max()), np.maximum(1e-4, 4 * dev))
        report[cp] = (err, bar)
        print(f"[{tag}] iteration {cp}: Linf vs the float64 trajectory per sample {err}, the reference's own fp32 deviation {g[f'{tag}_f32dev_probe_it{cp}']}")
        assert (err <= bar).all(), (tag, cp, err, bar)
    return report


@pytest.mark.gpu
def test_config2_batch32_1000_iterations_vs_float64_reference(g_long):
    """BASELINE configs[1] at full size: B = 32 (the 5 fixture maps + 27 more ring maps), 256^2, 1000 iterations."""
    li = long_inputs("cfg2")
    sos = np.concatenate([li["sos"], ring_sos_batch(256, 27, seed=21)])
    s = _solver()
    s.set_domain_size(256, source_location=li["loc"])
    trace, fields = _run_with_checkpoints(s, torch.from_numpy(sos).to(DEV), (100, 300, 1000))
    assert trace.shape == (1000, 32)
    _check("cfg2", g_long, trace, fields, 5)
    # the other 27 maps behave like the reference's five: RMSE floor of the trained network ~1.3e-5 .. 2.5e-5 (an
    # occasional ring phantom is still on its way down after 1000 iterations)
    assert np.median(trace[-1]) < 3e-5 and trace[-1].min() > 5e-6 and trace[-1].max() < 1e-2, (trace[-1].min(), trace[-1].max())
    # the same maps solved alone give the same answer bit for bit (samples are independent)
    alone = s.forward(torch.from_numpy(sos[:2]).to(DEV), num_iterations=100, residuals="last")
    assert np.array_equal(alone["wavefields"][0].cpu().numpy(), fields[100][:2])


@pytest.mark.gpu
def test_config4_batch16_512_2000_iterations_vs_float64_reference(g_long):
    """BASELINE configs[3] at full size: set_domain_size(512), B = 16, 2000 iterations."""
    if "cfg4_rmse_f64" not in g_long:
        pytest.skip("cfg4 fixture not generated")
    li = long_inputs("cfg4")
    sos = np.concatenate([li["sos"], ring_sos_batch(512, 15, seed=22)])
    s = _solver()
    s.set_domain_size(512, source_location=li["loc"])
    trace, fields = _run_with_checkpoints(s, torch.from_numpy(sos).to(DEV), (500, 1000, 2000))
    assert trace.shape == (2000, 16)
    # The reference itself is only metastable here: its residual creeps up from 4.7e-5 (iteration 700) to 6e-5 (1900),
    # with transient bursts, and its OWN fp32 run breaks away from its float64 run between iterations 1970 and 2000
    # (RMSE 6.9e-3 vs 6.3e-5, wavefields 0.72 apart -- recorded in the fixture, which makes the iteration-2000 wavefield
    # bar vacuous by construction).  So: early trace to 2 %, checkpoints 500 / 1000 within twice the reference's own
    # fp32 drift (2.9e-3 / 6.8e-3), the plateau level over iterations 1400 - 1600 to 10 %, everything finite.
    rep = _check("cfg4", g_long, trace, fields, 1, floor=slice(1400, 1600))
    assert rep[500][0][0] <= 2.9e-3 and rep[1000][0][0] <= 6.8e-3, rep
    assert np.isfinite(trace).all() and np.median(trace[1400:1600]) < 1e-4


@pytest.mark.gpu
def test_config5_fp16_unet_arc_source_convergence_to_tolerance(g_long):
    """BASELINE configs[4]: 512^2 transcranial map at the full 1.87x contrast, arc source map (support_functions.py:
    321-333), fp16 UNet / fp32 spectral residual, run until the residual RMSE is below a tolerance the reference's own
    fp32 run reaches -- against that run (fixture cfg5: RMSE trace of 3000 iterations, wavefield probes)."""
    if "cfg5_rmse_f32" not in g_long:
        pytest.skip("cfg5 fixture not generated")
    li = long_inputs("cfg5")
    ref = g_long["cfg5_rmse_f32"][:, 0]
    # the reference's trace falls steeply to ~3e-4 by iteration 300 and flattens out at 1.2e-4..1.3e-4 from iteration
    # ~1000 on (a property of the trained network on this out-of-distribution map); 2e-4 is crossed on the slope
    tol = 2e-4
    ref_cross = int(np.argmax(ref < tol)) + 1
    assert 300 < ref_cross < 700, ref_cross
    sos = torch.from_numpy(li["sos"]).to(DEV)
    out = {}
    for mode in ("fp32", "fp16"):
        s = _solver()
        s.set_unet_precision(mode)
        s.set_domain_size(512, source_map=torch.from_numpy(li["src_map"]).to(DEV))
        o = s.solve_to_tolerance(sos, tol=tol, max_iterations=3000, check_every=50)
        assert o["converged"], (mode, float(o["residual_norms"][-1].max()), tol)
        assert s.engine().unet_precision == mode
        out[mode] = o
        # iterations to tolerance: the reference's count rounded up to the check interval, within 15 %
        assert abs(o["iterations"] - ref_cross) <= max(50, 0.15 * ref_cross), (mode, o["iterations"], ref_cross)
        tr = o["residual_norms"][:, 0].cpu().numpy()
        k = min(len(tr), 300)
        bar = 2e-2 if mode == "fp32" else 5e-2
        assert np.abs(tr[:k] / ref[:k] - 1).max() <= bar, (mode, np.abs(tr[:k] / ref[:k] - 1).max())
    # the fp16 network converges to the fp32 answer: wavefield within 2e-3 of max after the same number of iterations
    n_it = min(out["fp32"]["iterations"], out["fp16"]["iterations"])
    a = _solver(); a.set_domain_size(512, source_map=torch.from_numpy(li["src_map"]).to(DEV))
    b = _solver(); b.set_unet_precision("fp16"); b.set_domain_size(512, source_map=torch.from_numpy(li["src_map"]).to(DEV))
    wa = a.forward(sos, num_iterations=1000, residuals="last")["wavefields"][0].cpu().numpy()
    wb = b.forward(sos, num_iterations=1000, residuals="last")["wavefields"][0].cpu().numpy()
    st = int(g_long["cfg5_stride"])
    gold = g_long["cfg5_wf_it1000"]
    scale = float(g_long["cfg5_wf_absmax_it1000"][0])
    e32 = np.abs(wa[:, :, ::st, ::st] - gold).max() / scale
    e16 = np.abs(wb[:, :, ::st, ::st] - gold).max() / scale
    assert e32 <= 1e-3 and e16 <= 5e-3, (e32, e16, n_it)


@pytest.mark.gpu
@pytest.mark.parametrize("rank", [0, 7])
def test_config3_shards_of_the_8_way_run(weights, rank):
    """BASELINE configs[2]: a batch of 256 random sound-speed maps sharded 8 ways -- on one GPU, the slices ranks 0 and 7 of the
    8-GPU run take (``shard_bounds(256, r, 8)``: 32 maps each; the maps are what ``bench.py --gpus 8`` gives rank r: random ring
    phantoms of the training distribution, dataloaders.py:115-156, seed r).
      * 40 iterations of the first 2 maps of the shard against the CPU oracle (L_inf(wavefield) <= 2e-4 * max|wf|, RMSE trace 2 %);
      * the full 1000 iterations at B = 32 through the size-independent properties: everything finite, the residual converged
        to the trained network's floor (median <= 1e-4, every map <= 1e-3 and a tenth of its start), and sample i of the shard equals sample i solved in a batch of 4 bit for bit
        (samples never interact, so the sharded run IS the unsharded one)."""
    from helmnet_amd.distributed import shard_batch, shard_bounds
    lo, hi = shard_bounds(256, rank, 8)
    assert (lo, hi) == (32 * rank, 32 * rank + 32)
    everything = torch.from_numpy(np.concatenate([ring_sos_batch(256, 32, seed=r) for r in range(8)]))     # the 256-map job
    sos = shard_batch(everything, rank, 8)
    assert sos.shape == (32, 1, 256, 256) and torch.equal(sos, torch.from_numpy(ring_sos_batch(256, 32, seed=rank)))
    s = _solver()
    s.set_domain_size(256, source_location=[30, 128])
    if rank == 0:   # the CPU oracle is the slow part of this test: one shard is compared against it, both go through the property checks
        out = s.forward(sos[:2].to(DEV), num_iterations=40, residuals="norms")
        want = O.solve(sos[:2], weights, O.point_source_map(256, [30, 128], 10.0), O.SpectralTables(256, 8, 2, 1.0), 40)
        err = float((out["wavefields"][0].cpu() - want["wavefield"]).abs().max())
        trace = torch.stack(want["trace"]).numpy()
        terr = float(np.abs(out["residual_norms"].cpu().numpy() / trace - 1).max())
        # the truth both fp32 runs approximate: the oracle in float64 on the same maps
        w64 = {k: v.double() for k, v in weights.items()}
        want64 = O.solve(sos[:2].double(), w64, O.point_source_map(256, [30, 128], 10.0).double(), O.SpectralTables(256, 8, 2, 1.0, dtype=torch.float64), 40)
        err64 = float((out["wavefields"][0].cpu().double() - want64["wavefield"]).abs().max())
        ora64 = float((want["wavefield"].double() - want64["wavefield"]).abs().max())
        scale = max(1.0, float(want["wavefield"].abs().max()))
        print(f"config3 rank {rank}: Linf(wf) after 40 it = {err:.3e} vs the fp32 oracle, {err64:.3e} vs the float64 oracle (the fp32 oracle itself: {ora64:.3e}); "
              f"|wf| max {scale:.3f}, trace rel err {terr:.3e}")
        # 40 iterations into the transient (|wf| ~ 2.7) two fp32 evaluations differ by ~1e-4 of the field's scale -- the CPU oracle
        # itself moves by that much with its batch size (oneDNN blocking): 1.0e-4 / 3.4e-4 absolute against the same HIP run at B = 4 / 2
        # (the reference's own fp32 run is 0.7 .. 8e-5 from its float64 run after 100 iterations).  The bar is on the
        # distance to the FLOAT64 evaluation: 2e-4 relative to the field's scale, or twice the fp32 oracle's own distance from it where that
        # is larger; two fp32 runs may then be up to the sum of their distances apart (r5: another fp32 summation order in the level-0
        # DoubleConvs moved the fp32-vs-fp32 figure from 3.4e-4 to 5.4e-4 with both equally close to float64)
        assert err64 <= max(2e-4 * scale, 2.0 * ora64) and terr <= 2e-2, (err, err64, ora64, terr)
        # fixed caps beside the relative bar: the distance to float64 whatever the fp32 oracle does (measured 8.8e-5), and the
        # fp32-vs-fp32 distance at its measured 5.4e-4 plus margin, so that a regression of the level-0 kernels still fails here
        assert err64 <= 4e-4 * scale and err <= 8e-4, (err, err64, scale)
    full = s.forward(sos.to(DEV), num_iterations=1000, residuals="norms")
    wf = full["wavefields"][0]
    rm = full["residual_norms"].cpu().numpy()
    print(f"config3 rank {rank}: RMSE it 1 max {rm[0].max():.3e}, it 1000 median {np.median(rm[-1]):.3e} max {rm[-1].max():.3e}")
    assert torch.isfinite(wf).all() and np.isfinite(rm).all()
    # the trained network's floor on its own distribution: 2e-5 median; single maps (a thick fast ring next to the source) stall near 4e-4
    assert np.median(rm[-1]) <= 1e-4 and rm[-1].max() <= 1e-3 and rm[-1].max() <= 0.1 * rm[0].max(), (rm[0].max(), np.median(rm[-1]), rm[-1].max())
    part = s.forward(sos[8:12].to(DEV), num_iterations=1000, residuals="norms")
    assert torch.equal(part["wavefields"][0], wf[8:12])
    assert torch.allclose(part["residual_norms"], full["residual_norms"][:, 8:12], rtol=1e-5)   # per-sample sums are float atomics


@pytest.mark.gpu
def test_smooth_random_maps_of_the_survey_follow_the_oracle_while_they_diverge(weights):
    """SURVEY 8(d) proposed `1 + U(0,1)` box-blurred maps for configs[2] (phantoms.smooth_random_sos).  Averaging 64 uniform
    samples gives c ~ 1.5 +- 0.04 EVERYWHERE -- source, PML and background included -- which the network, trained on a c = 1
    background, was never shown: the reference's iteration DIVERGES on them (|wavefield| ~ 2.6e3 after 40 iterations in the oracle
    and here alike).  The HIP path must still follow the oracle for as long as that is meaningful: 10 iterations, relative 1e-4."""
    from helmnet_amd.phantoms import smooth_random_sos
    sos = torch.from_numpy(smooth_random_sos(256, 4, seed=1))
    assert float(sos.min()) >= 1.0 and float(sos.max()) <= 2.0 and abs(float(sos.mean()) - 1.5) < 0.01
    s = _solver()
    s.set_domain_size(256, source_location=[30, 128])
    out = s.forward(sos.to(DEV), num_iterations=10, residuals="norms")
    want = O.solve(sos, weights, O.point_source_map(256, [30, 128], 10.0), O.SpectralTables(256, 8, 2, 1.0), 10)
    scale = float(want["wavefield"].abs().max())
    assert (out["wavefields"][0].cpu() - want["wavefield"]).abs().max() <= 1e-4 * scale
    trace = torch.stack(want["trace"]).numpy()
    assert np.abs(out["residual_norms"].cpu().numpy() / trace - 1).max() <= 1e-3
    assert trace[-1].min() > trace[0].max()       # the residual grows: this input is outside what the network can solve
