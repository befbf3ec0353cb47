"""garmentdreamer_amd -- MI355X-native (gfx950) implementation of GarmentDreamer's per-iteration
hot path: differentiable Gaussian rasterization + Score-Distillation guidance.

Layout (only what the path needs):
    csrc/                          hand-written HIP kernels + C-ABI (include/gd_raster.h)
    diff_gaussian_rasterization/   drop-in mirror of the reference's Python op
    cameras.py, gaussian_renderer.py, scene.py   host-side callers of the op (reference: GS/)
    guidance/                      threestudio-shaped SDS / VSD guidance (reference: TS/, NETF/)
    sds_loop.py, dist.py           the iteration that drives the path, view-sharded over RCCL
"""
__version__ = "0.1.0"

from . import _runtime_env as _runtime_env  # noqa: E402

_runtime_env.configure()   # before the first HIP call where possible (see the module docstring)
