"""CPU-only checks of the host logic and of the C-ABI shared library (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    """libhelmnet_hip.so loads without a GPU and exports every function include/*.h declares."""
    from helmnet_amd import _lib
    from helmnet_amd.build import build
    build()
    lib = _lib.load()
    hdr = open(os.path.join(REPO, "include", "helmnet_hip.h")).read()
    declared = set(re.findall(r"\b(hn_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"hn_ctx"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.hn_abi_version() == _lib.ABI_VERSION == 7
    assert lib.hn_weight_count(8, 4, 2) == 48160
    assert lib.hn_weight_count(16, 4, 2) == 0
    # no GPU here: creating a context must fail cleanly with a message, not crash
    if not torch.cuda.is_available():
        ctx = ctypes.c_void_p()
        assert lib.hn_create(ctypes.byref(ctx), 0) < 0
        assert b"HIP device" in lib.hn_last_error(None) or lib.hn_last_error(None)


def test_python_enum_tables_match_the_header():
    """The ctypes side passes plain integers: its tables must be the header's enumerators (name -> value), all of them."""
    from helmnet_amd import _lib
    hdr = open(os.path.join(REPO, "include", "helmnet_hip.h")).read()
    enums = {name: int(val) for name, val in re.findall(r"\b(HN_[A-Z0-9_]+)\s*=\s*(-?\d+)", hdr)}
    opt = {k: v for k, v in _lib.HN_OPTION.items() if v < 100}      # supported knobs: HN_OPT_*; laboratory knobs (>= 100): HN_EXP_*
    exp = {k: v for k, v in _lib.HN_OPTION.items() if v >= 100}
    for table, prefix, rename in ((opt, "HN_OPT_", {}), (exp, "HN_EXP_", {}), (_lib.HN_ACT, "HN_ACT_", {}),
                                  (_lib.HN_PRECISION, "HN_PREC_", {"valu": "FP32_VALU"}), (_lib.HN_COUNTER, "HN_CNT_", {})):
        in_header = {k: v for k, v in enums.items() if k.startswith(prefix)}
        mine = {prefix + rename.get(k, k).upper(): v for k, v in table.items()}
        assert mine == in_header, (prefix, set(mine.items()) ^ set(in_header.items()))
    assert int(re.search(r"#define\s+HN_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.ABI_VERSION


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under helmnet_amd/ may reference it."""
    for root, _, files in os.walk(os.path.join(REPO, "helmnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f


def test_weight_blob_order(weights_np):
    from helmnet_amd.engine import pack_weights, weight_names
    names = weight_names(4)
    assert len(names) == 88 and names[0] == "inc.double_conv.0.weight" and names[-1] == "outc.conv.bias"
    assert list(weights_np.keys()) == names          # exported in state_dict order
    blob = pack_weights(weights_np)
    assert blob.shape == (48160,) and blob.dtype == np.float32
    assert blob[0] == weights_np["inc.double_conv.0.weight"].reshape(-1)[0]
    relu = {k: v for k, v in weights_np.items() if not k.endswith("double_conv.1.weight")}
    b2 = pack_weights(relu, activation="relu")
    assert b2.shape == (48160,) and b2[432 + 8] == 0.0
    with pytest.raises(KeyError):
        pack_weights(relu, activation="prelu")


@pytest.mark.parametrize("n", [96, 256, 512])
def test_host_tables_match_reference_golden(n, g_setup):
    from helmnet_amd import FastLaplacianWithPML
    L = FastLaplacianWithPML(n, 8, 1.0, 2)
    assert np.array_equal(L.kx[0, 0, :, 1].numpy(), g_setup[f"n{n}_kx_row"])
    assert np.array_equal(L.ky[0, :, 0, 1].numpy(), g_setup[f"n{n}_ky_col"])
    assert np.array_equal(L.kx_sq[0, 0, :, 0].numpy(), g_setup[f"n{n}_kxsq_row"])
    assert np.array_equal(L.ky_sq[0, :, 0, 0].numpy(), g_setup[f"n{n}_kysq_col"])
    for nm in ("ax", "bx"):
        assert np.array_equal(getattr(L, nm)[0, 0].numpy(), g_setup[f"n{n}_{nm}_row"])
        assert torch.equal(getattr(L, nm)[0, 0], getattr(L, nm)[0, n // 3])
    for nm in ("ay", "by"):
        assert np.array_equal(getattr(L, nm)[0, :, 0].numpy(), g_setup[f"n{n}_{nm}_col"])
    sx, sy = L.sigmas()
    assert np.array_equal(sx[0].numpy(), g_setup[f"n{n}_sigma_x_row"])
    assert np.array_equal(sy[:, 0].numpy(), g_setup[f"n{n}_sigma_y_col"])
    assert L.kx.shape == (1, n, n, 2) and (L.kx[..., 0] == 0).all()


def test_solver_construction_and_setup(g_setup, hparams):
    from helmnet_amd import IterativeSolver
    s = IterativeSolver.from_exported_weights()
    s.freeze()
    assert s.hparams.domain_size == 96 and s.hparams["PMLsize"] == 8 and s.hparams.activation_function == "prelu"
    assert sum(p.numel() for p in s.f.parameters()) == 48160
    assert not any(p.requires_grad for p in s.parameters())
    assert np.allclose(s.source.numpy(), g_setup["n96_source"], atol=2e-6)
    s.set_domain_size(256, source_location=[30, 128])
    assert s.source.shape == (1, 2, 256, 256) and s.sigmas.shape == (2, 256, 256)
    assert s.f.states_dimension == [256, 128, 64, 32] and s.f.total_state_length == 87040
    assert [e.domain_size for e in s.f.enc] == [256, 128, 64, 32]
    assert np.allclose(s.source[0, :, 30, 128].numpy(), g_setup["n256_source_peak"], atol=1e-5)
    smap = torch.zeros(1, 2, 256, 256)
    smap[0, 0, 30, 120:130] = 1
    s.set_domain_size(256, source_map=smap)
    assert torch.equal(s.source, smap)
    k_sq, wf = s.get_initials(torch.full((2, 1, 256, 256), 2.0))
    assert torch.all(k_sq == 0.25) and wf.shape == (2, 2, 256, 256) and not wf.any()
    s.f.clear_states(wf)
    flat = s.f.get_states(flatten=True)
    assert flat.shape == (2, 2, 87040)
    back = s.f.unflatten_state(torch.arange(2 * 2 * 87040, dtype=torch.float32).view(2, 2, 87040))
    assert [tuple(t.shape) for t in back] == [(2, 2, 256, 256), (2, 2, 128, 128), (2, 2, 64, 64), (2, 2, 32, 32)]
    assert back[1][1, 1, 0, 0] == 3 * 87040 + 65536
    r = torch.ones(3, 2, 4, 4) * 2
    assert torch.allclose(IterativeSolver.test_loss_function(r), torch.full((3,), 2.0))
    with pytest.raises(NotImplementedError):
        IterativeSolver(**{**hparams, "architecture": "resnet"})
    with pytest.raises(NotImplementedError):
        IterativeSolver(**{**hparams, "activation_function": "relu_batchnorm"})
    assert IterativeSolver(**{**hparams, "activation_function": "gelu"}).f.activation_function == "gelu"
    with pytest.raises(RuntimeError):     # no CPU compute path
        s.get_residual(wf, k_sq)


def test_lightning_checkpoint_reader(tmp_path, weights, hparams):
    """A Lightning-format file (AttributeDict hyper_parameters, prefixed state_dict, legacy extra
    keys) loads with strict=False, as the reference's callers do."""
    from helmnet_amd import IterativeSolver
    from helmnet_amd.checkpoint import AttributeDict
    sd = {"f." + k: v for k, v in weights.items()}
    sd["source"] = torch.zeros(1, 2, 96, 96)
    sd["Lap.gamma_x"] = torch.zeros(1, 96, 96, 2)      # legacy tensor present in the shipped file
    hp = {k: v for k, v in hparams.items() if k != "test_data_path"}
    path = tmp_path / "toy.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": hp, "epoch": 1}, path)
    s = IterativeSolver.load_from_checkpoint(str(path), strict=False, test_data_path=None)
    assert isinstance(s.hparams, AttributeDict) and s.hparams.max_iterations == 1000
    assert torch.equal(s.f.inc.double_conv[0].weight, weights["inc.double_conv.0.weight"])
    with pytest.raises(RuntimeError):
        IterativeSolver.load_from_checkpoint(str(path), strict=True, test_data_path=None)
    ref = "/root/reference/trained_models/jcp_paper_trained_weights.ckpt"
    if os.path.exists(ref):
        s2 = IterativeSolver.load_from_checkpoint(ref, strict=False, test_data_path=None)
        for k, v in weights.items():
            assert torch.equal(s2.f.state_dict()[k], v), k


def test_phantoms():
    from helmnet_amd.phantoms import readme_sos, ring_sos_batch, smooth_random_sos
    a = ring_sos_batch(128, 4, seed=3)
    assert a.shape == (4, 1, 128, 128) and a.dtype == np.float32
    assert a.min() == 1.0 and 1.5 <= a.max() <= 2.0
    assert np.array_equal(a, ring_sos_batch(128, 4, seed=3))
    assert all(0.005 < (m > 1).mean() < 0.25 for m in a)
    r = readme_sos()
    assert r[0, 0, 100, 30] == 2.0 and r[0, 0, 169, 239] == 1.0 and r[0, 0, 0, 0] == 1.0
    s = smooth_random_sos(64, 2)
    assert s.shape == (2, 1, 64, 64) and 1.0 <= s.min() and s.max() <= 2.0


def test_gmres_hessenberg_least_squares_on_the_host_matches_lstsq():
    """helmnet_amd/gmres.py solves the (m + 1) x m Hessenberg least-squares problem once per restart cycle with Givens rotations (the
    reference leaves this to MATLAB's gmres, matlab/spectral_gmres_solver.m:107): the truncated solutions and the residual norm after every
    inner iteration equal numpy's dense least-squares answers."""
    import numpy as np
    from helmnet_amd.gmres import _back_substitute, _hessenberg_least_squares
    rng = np.random.default_rng(0)
    B, m = 3, 9
    H = np.zeros((B, m + 1, m), complex)
    for j in range(m):
        H[:, : j + 2, j] = rng.standard_normal((B, j + 2)) + 1j * rng.standard_normal((B, j + 2))
    beta = np.abs(rng.standard_normal(B)) + 1.0
    R, g, res = _hessenberg_least_squares(H, beta)
    for k in (1, 4, 9):
        y = _back_substitute(R, g, k)
        for b in range(B):
            e = np.zeros(k + 1, complex)
            e[0] = beta[b]
            yl = np.linalg.lstsq(H[b, : k + 1, :k], e, rcond=None)[0]
            assert np.allclose(y[b], yl) and np.isclose(res[b, k - 1], np.linalg.norm(e - H[b, : k + 1, :k] @ yl))
