"""helmnet_amd -- MI355X-native (gfx950) implementation of the helmnet IterativeSolver
inference loop: fused HIP kernels behind a C ABI (libhelmnet_hip.so, include/helmnet_hip.h),
with a host-side mirror of the reference's Python interface.

    from helmnet_amd import IterativeSolver
    solver = IterativeSolver.load_from_checkpoint("jcp_paper_trained_weights.ckpt", strict=False, test_data_path=None)
    solver.freeze(); solver.to("cuda:0")
    solver.set_domain_size(256, source_location=[30, 128])
    out = solver.forward(sos_maps, num_iterations=1000)

Importing the package does not need a GPU; computing does, and there is no CPU fallback.
"""
from .checkpoint import AttributeDict, read_lightning_checkpoint  # noqa: F401
from .laplacian import FastLaplacianWithPML  # noqa: F401
from .solver import IterativeSolver  # noqa: F401
from .source import SourceModule  # noqa: F401
from .unet import DoubleConv, EncoderBlock, HybridNet, OutConv, getActivationFunction  # noqa: F401

__all__ = [
    "AttributeDict",
    "DoubleConv",
    "EncoderBlock",
    "FastLaplacianWithPML",
    "HybridNet",
    "IterativeSolver",
    "OutConv",
    "SourceModule",
    "getActivationFunction",
    "read_lightning_checkpoint",
]
