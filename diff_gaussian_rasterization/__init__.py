"""Drop-in import name: ``from diff_gaussian_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`` (Garment_3DGS/gaussiansplatting/gaussian_renderer/__init__.py:14) resolves to
the MI355X-native implementation when the repository root is on ``sys.path``."""
from garmentdreamer_amd.diff_gaussian_rasterization import (  # noqa: F401
    BatchedRasterizationSettings,
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _C,
    _RasterizeGaussians,
    cpu_deep_copy_tuple,
    rasterize_gaussians,
    rasterize_gaussians_batched,
)
