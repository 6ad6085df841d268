import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def weights_np():
    with np.load(os.path.join(GOLDEN, "jcp_weights.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def weights(weights_np):
    return {k: torch.from_numpy(v) for k, v in weights_np.items()}


@pytest.fixture(scope="session")
def hparams():
    import json
    with open(os.path.join(GOLDEN, "jcp_hparams.json")) as f:
        return json.load(f)


def _load(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def g_setup():
    return _load("setup.npz")


@pytest.fixture(scope="session")
def g_teacher():
    return _load("teacher_forced.npz")


@pytest.fixture(scope="session")
def g_free():
    return _load("free_run.npz")
