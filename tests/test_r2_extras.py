"""Rows next to the hot path pinned by REFERENCE-generated fixtures (tests/golden/r2_extras.npz, made by
tests/golden/make_golden_r2.py), the classical baseline against a float64 direct solve, the untested activation
branches, and the real engine under a process group."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from golden_inputs import metric_inputs, teacher_inputs
from oracle import helmnet_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def g_r2():
    with np.load(os.path.join(REPO, "tests", "golden", "r2_extras.npz")) as z:
        return {k: z[k] for k in z.files}


def _c(a):
    return torch.complex(torch.from_numpy(a[:, 0].copy()), torch.from_numpy(a[:, 1].copy()))


# ---------------------------------------------------------------------------------------------- CPU
def test_metrics_vs_reference_fixture(g_r2):
    """helmnet_amd.metrics against the outputs of support_functions.py:10-48,124-130 on the same seeded fields."""
    from helmnet_amd.metrics import difference_to_reference, last_frame_difference, normalize_wavefield
    mi = metric_inputs()
    sample, ref, mask = _c(mi["sample"]), _c(mi["reference"]), torch.from_numpy(mi["mask"])
    assert np.allclose(normalize_wavefield(sample, [82, 48]).numpy(), g_r2["norm3d"], rtol=1e-6, atol=1e-7)
    assert np.allclose(normalize_wavefield(sample[1], [82, 48]).numpy(), g_r2["norm2d"], rtol=1e-6, atol=1e-7)
    d, s, r = difference_to_reference(sample, ref)
    assert d.shape == g_r2["diff"].shape == (3, 76, 76)
    assert np.allclose(d.numpy(), g_r2["diff"], rtol=1e-5, atol=1e-6)
    assert np.allclose(s.resolve_conj().numpy(), g_r2["diff_sample"], rtol=1e-6, atol=1e-7)
    assert np.allclose(r.resolve_conj().numpy(), g_r2["diff_reference"], rtol=1e-6, atol=1e-7)
    dm, _, _ = difference_to_reference(sample, ref, mask=mask)
    assert np.allclose(dm.numpy(), g_r2["diff_masked"], rtol=1e-5, atol=1e-6)
    li, rm = last_frame_difference(torch.from_numpy(mi["stream"]), ref)
    assert np.allclose(li.numpy(), g_r2["lfd_linf"], rtol=1e-5) and np.allclose(rm.numpy(), g_r2["lfd_rmse"], rtol=1e-5)


def test_smoothed_source_vs_reference_fixture(g_r2):
    """SourceModule(smooth=True): the host mirror and the oracle against source_module.py:41-79,94-116."""
    from helmnet_amd.source import SourceModule
    for n, loc in ((96, [82, 48]), (256, [30, 128])):
        m = SourceModule(image_size=n, omega=1, location=loc, amplitude=10, phase=0.3, smooth=True).spatial_map(0)
        want = g_r2[f"smooth{n}"]
        got = m.numpy() if n == 96 else m[:, ::4, ::4].numpy()
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= 2e-6 * scale
        assert abs(float(m.double().sum()) - float(g_r2[f"smooth{n}_sum"])) <= 1e-5 * abs(float(g_r2[f"smooth{n}_sum"]))
        assert np.allclose(m[0, loc[0], loc[1]].numpy(), g_r2[f"smooth{n}_peak"], rtol=1e-6)
        o = O.point_source_map(n, loc, 10.0, phase=0.3, smooth=True).permute(0, 2, 3, 1)
        assert (o - m).abs().max().item() <= 2e-6 * scale
    # the smoothing spreads the point: a genuinely different map from the unsmoothed one
    sharp = SourceModule(image_size=96, omega=1, location=[82, 48], amplitude=10, phase=0.3, smooth=False).spatial_map(0)
    assert (sharp - torch.from_numpy(g_r2["smooth96"])).abs().max().item() > 1.0


ACTS = ["relu", "leakyrelu", "celu", "tanh", "gelu", "tanhshrink", "softplus"]


@pytest.mark.parametrize("act", ACTS)
def test_oracle_parameter_free_activations_vs_reference_fixture(g_r2, weights, act):
    """architectures.py:20-41: the oracle's activation() against the reference network built with each name."""
    ti = teacher_inputs(64, 2, seed=4242)
    x = torch.from_numpy(g_r2["act_input"])
    d, st = O.unet_forward(x, O.unflatten_states(torch.from_numpy(ti["states"]), 64, 4), weights, act=act)
    assert (d - torch.from_numpy(g_r2[f"{act}_d"])).abs().max().item() <= 1e-5 * np.abs(g_r2[f"{act}_d"]).max()
    assert (O.flatten_states(st) - torch.from_numpy(g_r2[f"{act}_states"])).abs().max().item() <= 1e-5 * np.abs(g_r2[f"{act}_states"]).max()


@pytest.fixture(scope="module")
def g_sd():
    with np.load(os.path.join(REPO, "tests", "golden", "r2_state_depth.npz")) as z:
        return {k: z[k] for k in z.files}


def _sd_weights(g_sd):
    return {k[3:]: torch.from_numpy(v) for k, v in g_sd.items() if k.startswith("sd_")}


def _sd_bounds(n=64, depth=4):
    o, b = 0, []
    for d in range(depth):
        b.append((o, o + (n >> d) ** 2))
        o += (n >> d) ** 2
    return b


def test_oracle_state_depth_below_depth_vs_reference_fixture(g_sd):
    """architectures.py:203,212,250-251 (EncoderBlock without state) on a random depth-4 / state_depth-2 network: two
    consecutive reference passes; levels 2, 3 keep the (non-zero) values their slots were preset to."""
    w = _sd_weights(g_sd)
    assert w["enc.2.conv_signal.double_conv.0.weight"].shape[1] == 8 and "enc.2.conv_state.double_conv.0.weight" not in w
    st = O.unflatten_states(torch.from_numpy(g_sd["st0"]), 64, 4)
    for x, want_d, want_s in ((g_sd["x1"], g_sd["d1"], g_sd["s1"]), (g_sd["x2"], g_sd["d2"], g_sd["s2"])):
        d, st = O.unet_forward(torch.from_numpy(x), st, w, state_depth=2)
        assert (d - torch.from_numpy(want_d)).abs().max().item() <= 1e-5 * np.abs(want_d).max()
        assert (O.flatten_states(st) - torch.from_numpy(want_s)).abs().max().item() <= 1e-5 * np.abs(want_s).max()
    a = _sd_bounds()[2][0]
    assert np.array_equal(g_sd["s2"][:, :, a:], g_sd["st0"][:, :, a:])   # the reference never touched the stateless slots


def test_stateless_levels_are_packed_as_their_exact_stateful_equivalent(g_sd):
    from helmnet_amd.engine import pack_weights, weight_names
    w = _sd_weights(g_sd)
    blob = pack_weights(w, 4, "prelu", state_depth=2)
    full = pack_weights({**{k: torch.zeros(s) for k, s in (("enc.%d.conv_state.double_conv.%s" % (d, t), sh) for d in (2, 3) for t, sh in
                                                          (("0.weight", (2, 10, 3, 3)), ("0.bias", (2,)), ("1.weight", (1,)), ("2.weight", (2, 2, 3, 3)), ("2.bias", (2,))))},
                         **{k: (torch.cat([v, torch.zeros(8, 2, 3, 3)], 1) if k in ("enc.2.conv_signal.double_conv.0.weight", "enc.3.conv_signal.double_conv.0.weight") else v)
                            for k, v in w.items()}}, 4, "prelu")
    assert blob.shape == full.shape
    assert len(weight_names(4)) == 88
    diff = np.flatnonzero(blob != full)
    assert diff.size == 2   # only the two (irrelevant) PReLU slopes of the all-zero conv_state stand-ins differ (0.25 vs 0)


def test_explicit_operator_is_the_oracle_operator():
    """The assembled float64 system matrix (matlab/spectral_gmres_solver.m:50-90 construction) applies the same
    operator as the FFT formulation of spectral.py:31-79."""
    for n in (32, 48):
        rng = np.random.default_rng(n)
        sos = (1 + rng.random((n, n))).astype(np.float32)
        u = rng.standard_normal((1, 2, n, n)).astype(np.float32)
        t = O.SpectralTables(n, 8, 2, 1.0)
        r = O.get_residual(torch.from_numpy(u), torch.from_numpy((1 / sos) ** 2)[None, None], torch.zeros(1, 2, n, n), t)
        mat = O.assemble_helmholtz_matrix((1 / sos.astype(np.float64)) ** 2, 8, 2, 1.0)
        w = (mat @ (u[0, 0].astype(np.float64) + 1j * u[0, 1]).reshape(-1)).reshape(n, n)
        scale = np.abs(w).max()
        assert np.abs(w.real - r[0, 0].numpy()).max() <= 1e-5 * scale and np.abs(w.imag - r[0, 1].numpy()).max() <= 1e-5 * scale


def test_empty_shard_raises_instead_of_hanging():
    """fewer maps than ranks must fail on every rank up front (exercised without a process group through the
    same predicate)."""
    from helmnet_amd.distributed import shard_bounds
    assert shard_bounds(2, 2, 4) == (2, 2)   # rank 2 of 4 would own nothing


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("act", ACTS)
def test_gpu_parameter_free_activations_vs_reference_fixture(g_r2, weights, act):
    from helmnet_amd import HybridNet
    net = HybridNet(act, 4, 64, 8, 6, 2, 4)
    missing = net.load_state_dict(weights, strict=False)
    assert not missing.missing_keys
    net.to(DEV)
    ti = teacher_inputs(64, 2, seed=4242)
    net.set_states(torch.from_numpy(ti["states"]).to(DEV), flatten=True)
    d = net(torch.from_numpy(g_r2["act_input"]).to(DEV)).cpu().numpy()
    assert np.abs(d - g_r2[f"{act}_d"]).max() <= 1e-5 * np.abs(g_r2[f"{act}_d"]).max()
    assert np.abs(net.get_states(flatten=True).cpu().numpy() - g_r2[f"{act}_states"]).max() <= 1e-5 * np.abs(g_r2[f"{act}_states"]).max()


@pytest.mark.gpu
def test_gpu_state_depth_below_depth_vs_reference_fixture(g_sd):
    """HybridNet(state_depth=2) through the C ABI (the stateless levels run as their zero-padded stateful equivalent,
    engine.pack_weights) against the reference's two passes."""
    from helmnet_amd import HybridNet
    net = HybridNet("prelu", 4, 64, 8, 6, 2, 2)
    res = net.load_state_dict(_sd_weights(g_sd), strict=True)
    assert not res.missing_keys and not res.unexpected_keys and not hasattr(net.enc[2], "conv_state")
    net.to(DEV)
    net.set_states(torch.from_numpy(g_sd["st0"]).to(DEV), flatten=True)
    for x, want_d, want_s in ((g_sd["x1"], g_sd["d1"], g_sd["s1"]), (g_sd["x2"], g_sd["d2"], g_sd["s2"])):
        d = net(torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert np.abs(d - want_d).max() <= 1e-5 * np.abs(want_d).max()
        got_s = net.get_states(flatten=True).cpu().numpy()
        assert np.abs(got_s - want_s).max() <= 1e-5 * np.abs(want_s).max()
    a = _sd_bounds()[2][0]
    assert np.array_equal(got_s[:, :, a:], g_sd["st0"][:, :, a:])
    # a stateless slot holding NaN / Inf (e.g. left by a diverged run) is never read by the reference; here it must neither
    # reach the output (0 * NaN) nor be overwritten
    st = torch.from_numpy(g_sd["st0"]).clone()
    st[:, :, a:] = float("nan")
    st[0, 0, a + 5] = float("inf")
    net.set_states(st.to(DEV), flatten=True)
    d = net(torch.from_numpy(g_sd["x1"]).to(DEV)).cpu().numpy()
    assert np.abs(d - g_sd["d1"]).max() <= 1e-5 * np.abs(g_sd["d1"]).max()
    after = net.get_states(flatten=True).cpu()
    assert torch.isnan(after[:, :, a:]).sum() == st[:, :, a:].numel() - 1 and torch.isinf(after[0, 0, a + 5])
    assert np.abs(after[:, :, :a].numpy() - g_sd["s1"][:, :, :a]).max() <= 1e-5 * np.abs(g_sd["s1"]).max()


@pytest.mark.gpu
def test_gpu_smooth_activation_at_256_strip_and_deep_kernels(g_r2, weights):
    """tanh at 256^2: the GEN instances of the strip DoubleConv kernels and of the fused deep-level kernel."""
    from helmnet_amd import HybridNet
    net = HybridNet("tanh", 4, 256, 8, 6, 2, 4)
    net.load_state_dict(weights, strict=False)
    net.to(DEV)
    ti = teacher_inputs(256, 1, seed=4343)
    from helmnet_amd.laplacian import FastLaplacianWithPML
    sx, sy = FastLaplacianWithPML(domain_size=256, PMLsize=8, k=1.0, sigma_max=2.0).sigmas()
    x = torch.cat([torch.from_numpy(ti["wf"]), 1e3 * torch.from_numpy(ti["res"]), torch.stack([sx, sy]).float().unsqueeze(0)], 1)
    net.set_states(torch.from_numpy(ti["states"]).to(DEV), flatten=True)
    d = net(x.to(DEV)).cpu().numpy()
    assert np.abs(d[:, :, 3::7, 5::11] - g_r2["tanh256_d"]).max() <= 1e-5 * float(g_r2["tanh256_d_absmax"])
    st = net.get_states(flatten=True).cpu().numpy()[:, :, 1::37]
    assert np.abs(st - g_r2["tanh256_states"]).max() <= 1e-5 * np.abs(g_r2["tanh256_states"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n,loc", [(32, [12, 16]), (48, [14, 24])])
def test_gmres_on_hip_operator_vs_float64_direct_solve(n, loc):
    """Restarted GMRES on hn_residual against numpy.linalg.solve of the explicitly assembled operator
    (spectral_gmres_solver.m:50-107 is the reference's way of obtaining the classical answer)."""
    from helmnet_amd import IterativeSolver
    from helmnet_amd.gmres import gmres
    from helmnet_amd.phantoms import ring_sos_batch
    s = IterativeSolver.from_exported_weights(); s.freeze(); s.to(DEV)
    s.set_domain_size(n, source_location=loc)
    sos = ring_sos_batch(n, 2, seed=n)
    out = gmres(s, torch.from_numpy(sos).to(DEV), restart=40, max_outer=60, tol=2e-6)
    got = out["wavefield"].cpu().numpy()
    src = s.source.detach().cpu().numpy()[0]
    for b in range(2):
        want = O.direct_solve(sos[b, 0], src, 8, 2.0, 1.0)
        assert np.abs(got[b] - want).max() <= 2e-4 * np.abs(want).max(), (b, np.abs(got[b] - want).max(), np.abs(want).max())


_PG_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # before any other GPU work in this process
from helmnet_amd import IterativeSolver
from helmnet_amd.distributed import allreduce_residual_norms, solve_sharded
from helmnet_amd.phantoms import ring_sos_batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to(dev)
s.set_domain_size(128, source_location=[20, 64])
sos = torch.from_numpy(ring_sos_batch(128, 3, seed=5)).to(dev)
plain = s.forward(sos, num_iterations=60, residuals="norms")
state = {}
def solve(local, n_iter):
    if "wf" not in state:
        o = s.forward(local, num_iterations=n_iter, residuals="norms")
        state["k_sq"] = s.get_initials(local)[0].contiguous()
    else:
        o = s.n_steps(state["wf"], state["k_sq"], state["res"], n_iter, residuals="norms")
    state["wf"], state["res"] = o["wavefields"][0], o["last_residual"]
    return {"wavefield": state["wf"], "rmse": o["residual_norms"][-1]}
r = solve_sharded(solve, sos, 60, tol=None, gather=True)
assert torch.equal(r["wavefield_all"], plain["wavefields"][0]), "sharded solve differs from forward()"
assert torch.allclose(r["worst_rmse"].cpu(), plain["residual_norms"][-1].max().reshape(1).cpu(), rtol=1e-5)   # per-sample sums are float atomics
t = s.solve_to_tolerance(sos, tol=1e-3, max_iterations=200, check_every=20, norm_reduce=allreduce_residual_norms)
u = s.solve_to_tolerance(sos, tol=1e-3, max_iterations=200, check_every=20)
assert t["converged"] and t["iterations"] == u["iterations"] and torch.equal(t["wavefield"], u["wavefield"])
dist.barrier(); dist.destroy_process_group()
print("PG_OK", r["iterations"], t["iterations"])
'''


@pytest.mark.gpu
def test_real_engine_under_an_rccl_process_group_world_size_1(tmp_path):
    """solve_sharded + solve_to_tolerance(norm_reduce=allreduce_residual_norms) on the HIP engine with
    init_process_group("nccl", world_size=1): the RCCL code path runs on hardware and equals the plain forward()."""
    script = tmp_path / "pg.py"
    script.write_text(_PG_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), REPO], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PG_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


_PG2_SCRIPT = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
from helmnet_amd import IterativeSolver
from helmnet_amd.distributed import shard_bounds, solve_sharded
from helmnet_amd.phantoms import ring_sos_batch
s = IterativeSolver.from_exported_weights(); s.freeze(); s.to(dev)
s.set_domain_size(128, source_location=[20, 64])
sos = torch.from_numpy(ring_sos_batch(128, 5, seed=5)).to(dev)          # ragged: shards of 3 and 2
plain = s.forward(sos, num_iterations=40, residuals="norms")
state = {}
def solve(local, n_iter):
    if "wf" not in state:
        o = s.forward(local, num_iterations=n_iter, residuals="norms")
        state["k_sq"] = s.get_initials(local)[0].contiguous()
    else:
        o = s.n_steps(state["wf"], state["k_sq"], state["res"], n_iter, residuals="norms")
    state["wf"], state["res"] = o["wavefields"][0], o["last_residual"]
    return {"wavefield": state["wf"], "rmse": o["residual_norms"][-1]}
r = solve_sharded(solve, sos, 40, tol=None, gather=True)
lo, hi = shard_bounds(5, rank, world)
assert torch.equal(r["wavefield"], plain["wavefields"][0][lo:hi]), "a shard differs from the same samples of the unsharded batch"
if rank == 0:
    assert torch.equal(r["wavefield_all"], plain["wavefields"][0])
assert torch.allclose(r["worst_rmse"].cpu(), plain["residual_norms"][-1].max().reshape(1).cpu(), rtol=1e-5)
dist.barrier(); dist.destroy_process_group()
print("PG2_OK", rank)
'''


@pytest.mark.gpu
def test_two_rank_rccl_sharded_solve_on_two_devices(tmp_path):
    """BASELINE configs[2] in miniature on real hardware: two ranks, one GPU each, RCCL residual-norm all-reduce + gather; every shard equals the
    same samples of the unsharded batch bit for bit.  Skipped on boxes with one device (the gloo twin runs everywhere)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    import socket
    script = tmp_path / "pg2.py"
    script.write_text(_PG2_SCRIPT)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), REPO], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs) and all("PG2_OK" in o[0] for o in outs), [o[0][-500:] + o[1][-2000:] for o in outs]
