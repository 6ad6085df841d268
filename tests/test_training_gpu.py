"""Parity of the HIP training step (hn_train_grad / hn_adam_step / hn_residual_vjp, through the C ABI) against the
CPU oracle's autograd and against the reference-generated fixture ``tests/golden/train_step.npz`` (make_golden_train.py).
Needs a real MI355X: ``python -m pytest tests -m gpu``.

Tolerances (fp32; gradients are sums over up to ~10^6 terms, compared relative to the largest entry of each tensor):
  * adjoint operator vs autograd of the oracle's residual:           1e-5 * max
  * one unrolled iteration, every intermediate tensor and gradient:  1e-4 * max (forward tensors 1e-5 * max)
  * ten unrolled iterations vs the REFERENCE's autograd:             loss 1e-5 relative, gradients 1e-3 * max (the iteration
    amplifies rounding) -- and the same bar for the oracle, pinned on the CPU in test_training_cpu.py
"""
import os

import numpy as np
import pytest
import torch

from golden_inputs import teacher_inputs
from helmnet_amd.engine import pack_weights, unpack_weights, weight_names
from helmnet_amd.phantoms import ring_sos_batch
from oracle import helmnet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def solver():
    from helmnet_amd import IterativeSolver
    s = IterativeSolver.from_exported_weights()
    s.to(DEV)
    return s


def rel(got, want):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("n", [32, 64, 96, 112, 160, 256])
def test_residual_vjp_is_the_adjoint_of_the_residual(solver, n):
    """pow2 radix-4 (32, 64, 256), prime-factor (96 = 3 * 32, 160 = 5 * 32) and dense (112) paths."""
    solver.set_domain_size(n, source_location=[n // 3, n // 2])
    eng = solver.engine()
    ti = teacher_inputs(n, 2, seed=77 + n)
    wf = torch.from_numpy(ti["wf"]).requires_grad_(True)
    g = torch.from_numpy(teacher_inputs(n, 2, seed=78 + n)["wf"])
    k_sq = (1.0 / torch.from_numpy(ti["sos"])) ** 2
    t = O.SpectralTables(n, 8, 2, 1.0)
    res = O.get_residual(wf, k_sq, torch.zeros(1, 2, n, n), t)
    (want,) = torch.autograd.grad(res, wf, g)
    got = eng.residual_vjp(g.to(DEV), k_sq.to(DEV).contiguous())
    assert rel(got, want) <= 1e-5
    # <L u, g> == <u, L^H g> with the library's own forward operator
    fwd = eng.residual(wf.detach().to(DEV), k_sq.to(DEV).contiguous(), torch.zeros(1, 2, n, n, device=DEV))
    lhs = float((fwd.double() * g.to(DEV).double()).sum())
    rhs = float((wf.detach().to(DEV).double() * got.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0)


def _oracle_grads(weights, wf, res, st, k_sq, src, t, n_unroll, act="prelu", tape=None):
    w = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    wf, res, st = (x.clone().requires_grad_(True) for x in (wf, res, st))
    loss, wfs, ress, sts = O.training_loss(wf, res, st, k_sq, src, w, t, n_unroll, act=act, tape=tape)
    loss.backward()
    return loss.detach(), w, (wf.grad, res.grad, st.grad), (wfs, ress, sts)


def _blob(named, names):
    return torch.cat([named[k].reshape(-1) for k in names])


def _report(errs, bar):
    bad = {k: v for k, v in errs.items() if not v <= bar}
    assert not bad, f"above {bar}: {bad}\nall: {errs}"


@pytest.mark.parametrize("n,b", [(96, 2), (64, 3), (256, 1)])
def test_one_unrolled_iteration_matches_oracle_autograd_tensor_by_tensor(solver, weights, n, b):
    solver.set_domain_size(n, source_location=[n // 3, n // 2])
    eng = solver.engine()
    ti = teacher_inputs(n, b, seed=500 + n)
    wf, res, st, sos = (torch.from_numpy(ti[k]) for k in ("wf", "res", "states", "sos"))
    st = 0.2 * st
    k_sq = (1.0 / sos) ** 2
    t = O.SpectralTables(n, 8, 2, 1.0)
    src = O.point_source_map(n, [n // 3, n // 2], 10.0)
    tape = {}
    loss, w, gin, lists = _oracle_grads(weights, wf, res, st, k_sq, src, t, 1, tape=tape)
    names = weight_names(4)
    blob = torch.from_numpy(pack_weights({k: v.detach() for k, v in weights.items()})).to(DEV)
    out = eng.train_grad(blob, wf.to(DEV), res.to(DEV), st.to(DEV), k_sq.to(DEV).contiguous(), src.to(DEV), 1, 1e4, input_grads=True)
    torch.cuda.synchronize()
    # forward tape, level by level
    fwd = {}
    peek = {"x": "x{d}", "out": "out{d}", "u": "u{d}", "y": "y{d}", "sig_mid": "enc.{d}.conv_signal.mid", "st_mid": "enc.{d}.conv_state.mid",
            "dec_mid": "decode.{d}.mid"}
    for kind, pat in peek.items():
        for d in range(5):
            name = pat.format(d=d)
            if name in tape:
                fwd[f"{kind}{d}"] = rel(eng.train_peek(kind, d, b), tape[name])
    fwd["inc_mid"] = rel(eng.train_peek("inc_mid", 0, b), tape["inc.mid"])
    fwd["wf1"] = rel(out["wavefields"][0], lists[0][0])
    fwd["res1"] = rel(out["residuals"][0], lists[1][0])
    fwd["st1"] = rel(out["states"][0], lists[2][0])
    _report(fwd, 1e-5)
    assert abs(float(out["loss"][0]) - float(loss)) <= 1e-5 * float(loss)
    # activation gradients, then parameter gradients tensor by tensor
    bwd = {}
    for kind, pat in (("g_y", "y{d}"), ("g_u", "u{d}"), ("g_x", "x{d}"), ("g_out", "out{d}")):
        for d in range(5):
            name = pat.format(d=d)
            if name in tape and tape[name].grad is not None:
                bwd[f"{kind}{d}"] = rel(eng.train_peek(kind, d, b), tape[name].grad)
    bwd["grad_wf"] = rel(out["grad_wf"], gin[0])
    bwd["grad_res"] = rel(out["grad_res"], gin[1])
    bwd["grad_states"] = rel(out["grad_states"], gin[2])
    got = unpack_weights(out["grad"], 4)
    for k in names:
        bwd[k] = rel(torch.from_numpy(got[k]), w[k].grad)
    _report(bwd, 1e-4)


def test_smooth_activation_gradients(solver, weights):
    """tanh instead of PReLU (architectures.py:24-25): act'(z) in the epilogue, no slope parameter."""
    from helmnet_amd import IterativeSolver
    n, b = 32, 2
    s = IterativeSolver.from_exported_weights(activation_function="tanh")
    s.to(DEV)
    s.set_domain_size(n, source_location=[10, 16])
    eng = s.engine()
    ti = teacher_inputs(n, b, seed=901)
    wf, res, st, sos = (torch.from_numpy(ti[k]) for k in ("wf", "res", "states", "sos"))
    k_sq = (1.0 / sos) ** 2
    t = O.SpectralTables(n, 8, 2, 1.0)
    src = O.point_source_map(n, [10, 16], 10.0)
    wts = {k: v for k, v in weights.items() if not k.endswith("double_conv.1.weight")}
    loss, w, gin, _ = _oracle_grads(wts, wf, res, st, k_sq, src, t, 2, act="tanh")
    blob = torch.from_numpy(pack_weights(wts, 4, "tanh")).to(DEV)
    out = eng.train_grad(blob, wf.to(DEV), res.to(DEV), st.to(DEV), k_sq.to(DEV).contiguous(), src.to(DEV), 2, 1e4, input_grads=True)
    got = unpack_weights(out["grad"], 4)
    errs = {k: rel(torch.from_numpy(got[k]), w[k].grad) for k in wts}
    errs["grad_wf"] = rel(out["grad_wf"], gin[0])
    _report(errs, 2e-4)
    assert all(float(np.abs(got[k]).max()) == 0.0 for k in got if k.endswith("double_conv.1.weight"))
    assert abs(float(out["loss"][0]) - float(loss)) <= 1e-5 * float(loss)


@pytest.fixture(scope="module")
def g_train():
    with np.load(os.path.join(GOLDEN, "train_step.npz")) as z:
        return {k: z[k] for k in z.files}


def _fixture_inputs(g_train):
    n, b = 96, 2
    sos = torch.from_numpy(ring_sos_batch(n, b, seed=21))
    k_sq = (1.0 / sos) ** 2
    src = O.point_source_map(n, [82, 48], 10.0).repeat(b, 1, 1, 1)
    return n, b, k_sq, src, (torch.from_numpy(g_train[k]) for k in ("wf0", "res0", "st0"))


def test_ten_unrolled_iterations_match_the_reference_autograd(solver, weights, g_train):
    n, b, k_sq, src, (wf0, res0, st0) = _fixture_inputs(g_train)
    solver.set_domain_size(n, source_location=[82, 48])
    eng = solver.engine()
    blob = torch.from_numpy(pack_weights(weights)).to(DEV)
    out = eng.train_grad(blob, wf0.to(DEV), res0.to(DEV), st0.to(DEV), k_sq.to(DEV).contiguous(), src.to(DEV).contiguous(), 10, 1e4, input_grads=True)
    assert abs(float(out["loss"][0]) - float(g_train["loss"])) <= 1e-5 * float(g_train["loss"])
    errs = {"wf_T": rel(out["wavefields"][-1], torch.from_numpy(g_train["wf_T"])),
            "res_T": rel(out["residuals"][-1], torch.from_numpy(g_train["res_T"])),
            "st_T": rel(out["states"][-1], torch.from_numpy(g_train["st_T"]))}
    _report(errs, 1e-4)
    rmse = eng.rmse(out["residuals"][-1]).cpu().numpy()
    assert np.allclose(rmse, g_train["res_rmse"][-1], rtol=1e-4)
    want = unpack_weights(g_train["grad"], 4)
    got = unpack_weights(out["grad"], 4)
    gerr = {k: rel(torch.from_numpy(got[k]), torch.from_numpy(want[k])) for k in want}
    gerr["grad_wf0"] = rel(out["grad_wf"], torch.from_numpy(g_train["grad_wf0"]))
    gerr["grad_res0"] = rel(out["grad_res"], torch.from_numpy(g_train["grad_res0"]))
    gerr["grad_st0"] = rel(out["grad_states"], torch.from_numpy(g_train["grad_st0"]))
    _report(gerr, 1e-3)
    # whole-blob figure
    assert rel(out["grad"], torch.from_numpy(g_train["grad"])) <= 2e-4


def test_gradients_are_bit_reproducible_and_batch_independent(solver, weights, g_train):
    n, b, k_sq, src, (wf0, res0, st0) = _fixture_inputs(g_train)
    solver.set_domain_size(n, source_location=[82, 48])
    eng = solver.engine()
    blob = torch.from_numpy(pack_weights(weights)).to(DEV)
    args = [x.to(DEV).contiguous() for x in (wf0, res0, st0, k_sq, src)]
    a = eng.train_grad(blob, *args, 3, 1e4, input_grads=True)
    b2 = eng.train_grad(blob, *args, 3, 1e4, input_grads=True)
    assert torch.equal(a["grad"], b2["grad"]) and torch.equal(a["grad_wf"], b2["grad_wf"])
    # sample 1 alone: its input gradients are those it had inside the batch, scaled by the batch size (the loss is a mean)
    one = eng.train_grad(blob, *[x[1:2].contiguous() for x in args], 3, 1e4, input_grads=True)
    assert rel(one["grad_wf"][0] / 2, a["grad_wf"][1]) <= 1e-6
    assert torch.equal(one["wavefields"][-1][0], a["wavefields"][-1][1])


def test_adam_three_steps_match_the_reference_optimiser(solver, weights, g_train):
    n, b, k_sq, src, (wf0, res0, st0) = _fixture_inputs(g_train)
    solver.set_domain_size(n, source_location=[82, 48])
    eng = solver.engine()
    lr, b1, b2, eps, wd, clip = (float(v) for v in g_train["adam_hparams"])
    w0 = torch.from_numpy(pack_weights(weights))
    blob = w0.clone().to(DEV)
    m, v = torch.zeros_like(blob), torch.zeros_like(blob)
    args = [x.to(DEV).contiguous() for x in (wf0, res0, st0, k_sq, src)]
    losses = []
    for step in range(1, 4):
        out = eng.train_grad(blob, *args, 10, 1e4)
        losses.append(float(out["loss"][0]))
        eng.adam_step(blob, out["grad"], m, v, step, lr, (b1, b2), eps, wd, clip)
    assert np.allclose(losses, g_train["adam_losses"], rtol=2e-3), (losses, g_train["adam_losses"])
    want = torch.from_numpy(g_train["adam_weights"])
    delta_want, delta_got = want - w0, blob.cpu() - w0
    err = (delta_got - delta_want).abs()
    # Adam's normalised update m / sqrt(v) turns a relative gradient error into an absolute step error of about lr * that
    # error where |g| is tiny; the bulk must agree to a small fraction of one step (lr = 1e-3), everything to within one step
    assert float(err.quantile(0.99)) <= 0.05 * lr, float(err.quantile(0.99))
    assert float(err.max()) <= 1.5 * lr, float(err.max())
    assert float((delta_got * delta_want).sum() / (delta_want.norm() * delta_got.norm())) >= 0.999


def test_adam_kernel_matches_torch_adam_on_given_gradients(solver):
    eng = solver.engine()
    gen = torch.Generator().manual_seed(5)
    nW = 48160
    w0 = torch.randn(nW, generator=gen) * 0.1
    grads = [torch.randn(nW, generator=gen) * (10.0 ** torch.randint(-6, 1, (nW,), generator=gen).float()) for _ in range(5)]
    want = O.adam_reference(w0, grads, 1e-3, (0.9, 0.95), 1e-8, 1e-6, 1.0)
    blob, m, v = w0.clone().to(DEV), torch.zeros(nW, device=DEV), torch.zeros(nW, device=DEV)
    mask = torch.ones(nW, dtype=torch.uint8, device=DEV)
    mask[100:110] = 0
    for i, g in enumerate(grads):
        eng.adam_step(blob, g.to(DEV), m, v, i + 1, 1e-3, (0.9, 0.95), 1e-8, 1e-6, 1.0, trainable=mask)
    got = blob.cpu()
    assert torch.equal(got[100:110], w0[100:110])
    keep = torch.ones(nW, dtype=torch.bool)
    keep[100:110] = False
    assert float((got - want)[keep].abs().max()) <= 2e-6      # 5 steps of 1e-3: agreement to ~1e-4 of the total update


def test_trainer_runs_training_steps_and_learns(solver):
    """A few optimiser steps from a fresh (Xavier, gain 0.02) network on a tiny buffer: the loss of a fixed probe batch drops, the
    replay buffer is refilled as in training_step (hybridnet.py:436-463), and the trained weights reach the module."""
    from helmnet_amd import IterativeSolver
    from helmnet_amd.training import Trainer
    torch.manual_seed(0)
    np.random.seed(0)
    import random
    random.seed(0)
    n = 32
    s = IterativeSolver(domain_size=n, k=1.0, omega=1, PMLsize=8, sigma_max=2, source_location=[10, 16], activation_function="prelu",
                        batch_size=4, buffer_size=8, depth=4, features=8, learning_rate=1e-3, minimum_learning_rate=1e-4, weight_decay=1e-6,
                        gradient_clip_val=1, max_iterations=100, source_amplitude=10, state_channels=2, state_depth=4, unrolling_steps=4)
    s.to(DEV)
    tr = Trainer(s)
    sos = torch.from_numpy(ring_sos_batch(n, 8, seed=3))
    tr.fill_replay_buffer(sos)
    assert all(e is not None and e.iteration == 10 * i for i, e in enumerate(tr.replaybuffer.buffer))
    probe = tr.replaybuffer.sample(4)
    before = float(tr.loss_and_grad(*probe[:5])["loss"][0])
    w_before = tr.weights.clone()
    tr.current_epoch = 3        # maxiter = 61: advanced experiences may be kept
    for i in range(12):
        out = tr.training_step(sos[:4].to(DEV), i)
        assert np.isfinite(float(out["loss"]))
    after = float(tr.loss_and_grad(*probe[:5])["loss"][0])
    assert after < before, (before, after)
    assert not torch.equal(w_before, tr.weights)
    its = [e.iteration for e in tr.replaybuffer.buffer]
    assert any(0 < it < 61 and it % 10 != 0 for it in its) or tr.new_sos > 0
    mean = tr.training_epoch_end()
    assert np.isfinite(mean) and tr.current_epoch == 4
    # the module's parameters now hold the trained blob, and inference uses them
    sd = {k: v for k, v in s.f.state_dict().items()}
    assert np.array_equal(pack_weights(sd), tr.weights.cpu().numpy())
    out = s.forward(sos[:2].to(DEV), num_iterations=3)
    assert torch.isfinite(out["wavefields"][0]).all()
    # resume: a second trainer loaded from the first one's state continues bit-identically
    st = tr.state_dict()
    tr2 = Trainer(s)
    tr2.load_state_dict(st)
    a = tr.loss_and_grad(*probe[:5])
    tr.optimizer_step()
    b = tr2.loss_and_grad(*probe[:5])
    tr2.optimizer_step()
    assert torch.equal(a["grad"], b["grad"]) and torch.equal(tr.weights, tr2.weights)
