"""Per-view / batched render wrappers around the rasterizer op.

``render`` restates Garment_3DGS/gaussiansplatting/gaussian_renderer/__init__.py:18-103 for the one
live configuration of the pipeline (SH + scale/rotation path; ``convert_SHs_python`` and
``compute_cov3D_python`` are False: Garment_3DGS/gaussiansplatting/arguments/__init__.py:65-67) and
returns the same dict keys.  ``render_batch`` is the MI355X-side addition: V views of the same
Gaussians through ONE launch set (one binning sort, one host sync) instead of the Python loop at
Garment_3DGS/threestudio/systems/GaussianDreamer.py:189-191.
"""
from __future__ import annotations

import math

import torch

from .cameras import CameraBatch
from .diff_gaussian_rasterization import (BatchedRasterizationSettings, GaussianRasterizationSettings,
                                          GaussianRasterizer)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render one view.  ``bg_color`` must be on the GPU (as in the reference)."""
    xyz = pc.get_xyz
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    shs, colors_precomp = (pc.get_features, None) if override_color is None else (None, override_color)
    rendered_image, radii, depth, alpha = rasterizer(
        means3D=xyz.float(),
        means2D=screenspace_points.float(),
        shs=None if shs is None else shs.float(),
        colors_precomp=colors_precomp,
        opacities=pc.get_opacity.float(),
        scales=pc.get_scaling.float(),
        rotations=pc.get_rotation.float(),
        cov3D_precomp=None)

    return {"render": rendered_image,
            "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0,
            "radii": radii,
            "depth_3dgs": depth,
            "alpha": alpha}


def render_batch(cameras, pc, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, capacity=None):
    """Render ``len(cameras)`` views in one launch set.  Returns the per-view dict keys stacked on
    a leading V axis: render [V,3,H,W], viewspace_points [V,P,3] (gradient holder), radii [V,P],
    depth_3dgs [V,1,H,W], alpha [V,1,H,W].  ``capacity``: a ``diff_gaussian_rasterization._C.InstanceCapacity`` kept by the
    caller across iterations -- the forward pass then runs without its host synchronisation (see that class)."""
    xyz = pc.get_xyz
    cb = cameras if isinstance(cameras, CameraBatch) else CameraBatch(cameras, xyz.device)
    V = cb.viewmatrix.shape[0]
    screenspace_points = torch.zeros((V,) + tuple(xyz.shape), dtype=xyz.dtype, requires_grad=True,
                                     device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rs = BatchedRasterizationSettings(
        image_height=cb.image_height, image_width=cb.image_width, tanfovx=cb.tanfovx, tanfovy=cb.tanfovy,
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=cb.viewmatrix, projmatrix=cb.projmatrix,
        sh_degree=pc.active_sh_degree, campos=cb.campos, prefiltered=False, debug=False, capacity=capacity)
    rasterizer = GaussianRasterizer(raster_settings=rs)
    if hasattr(pc, "activated"):     # flat-buffer GaussianModel: all activations in one launch
        feats, opacities, scales, rotations = pc.activated()
    else:
        feats, opacities, scales, rotations = pc.get_features, pc.get_opacity, pc.get_scaling, pc.get_rotation
    shs, colors_precomp = (feats, None) if override_color is None else (None, override_color)
    rendered, radii, depth, alpha = rasterizer(
        means3D=xyz.float(), means2D=screenspace_points.float(), shs=None if shs is None else shs.float(),
        colors_precomp=colors_precomp, opacities=opacities.float(), scales=scales.float(),
        rotations=rotations.float(), cov3D_precomp=None)
    return {"render": rendered, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth_3dgs": depth, "alpha": alpha}
