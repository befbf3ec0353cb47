"""Second in-tree HIP library: kernels of the guidance step (include/gd_nn.h)."""
from __future__ import annotations

NN_SOURCES = [
    ("nn_groupnorm.hip", ["-munsafe-fp-atomics"]),
    ("nn_conv3x3.hip", []),
    ("nn_elementwise.hip", []),
    ("nn_attention.hip", []),
    ("nn_prologue.hip", ["-munsafe-fp-atomics"]),
    ("nn_fp8.hip", []),
    ("nn_linear.hip", []),
    ("nn_lora.hip", []),
    ("nn_gemm.hip", []),
]


def build(force: bool = False, verbose: bool = False):
    from . import _build
    return [_build.build_library("libgd_nn.so", NN_SOURCES, force, verbose)]
