"""MI355X-native ``diff_gaussian_rasterization`` -- same public surface as the reference package
(``DGR/diff_gaussian_rasterization/__init__.py``, DGR = Garment_3DGS/gaussiansplatting/submodules/
diff-gaussian-rasterization) so ``GS/gaussian_renderer/__init__.py:14,51,86-94`` works unchanged:

    GaussianRasterizationSettings   NamedTuple, same fields/order            (__init__.py:160-172)
    GaussianRasterizer              nn.Module: forward(...), markVisible(...)  (:174-223)
    rasterize_gaussians             functional form                            (:21-42)

Forward returns ``(color[3,H,W], radii[P] int32, depth[1,H,W], alpha[1,H,W])``; backward returns
gradients for ``(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
cov3Ds_precomp, None)`` -- the reference's orders (:98, :146-156).

Additions (no reference counterpart): ``BatchedRasterizationSettings`` /
``rasterize_gaussians_batched`` / ``GaussianRasterizer.forward_batched`` render V views of the same
Gaussians with one launch set and one host sync.
"""
from __future__ import annotations

from typing import NamedTuple, Sequence

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        # 19 positional arguments, in the native order (rasterize_points.h:18-38)
        args = (
            raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
            raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
            raster_settings.tanfovx, raster_settings.tanfovy, raster_settings.image_height,
            raster_settings.image_width, sh, raster_settings.sh_degree, raster_settings.campos,
            raster_settings.prefiltered, raster_settings.debug,
        )
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # snapshot before anything can be corrupted
            try:
                num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer = \
                    _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer = \
                _C.rasterize_gaussians(*args)

        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, alpha)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer, alpha) = ctx.saved_tensors

        # 24 positional arguments, R at index 19 (rasterize_points.h:40-65)
        args = (raster_settings.bg, means3D, radii, colors_precomp, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy, grad_color, grad_depth,
                grad_alpha, sh, raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered,
                binningBuffer, imgBuffer, alpha, raster_settings.debug)

        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
                 grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
             grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(*args)

        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class BatchedRasterizationSettings(NamedTuple):
    """V views of one scene: viewmatrix/projmatrix [V,4,4], campos [V,3], tanfov*: V floats."""
    image_height: int
    image_width: int
    tanfovx: Sequence[float]
    tanfovy: Sequence[float]
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    capacity: object = None      # _C.InstanceCapacity: forward pass without the host read-back of the instance count


class _RasterizeGaussiansBatched(torch.autograd.Function):
    """means2D is a ``[V,P,3]`` gradient holder (one screen-space gradient per view, as the
    densification statistics need: GaussianDreamer.py:270-276)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs):
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, list(rs.tanfovx), list(rs.tanfovy), rs.image_height, rs.image_width,
                sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        num_rendered, color, depth, alpha, radii, geomBuffer, binningBuffer, imgBuffer = \
            _C.rasterize_gaussians_batched(*args, capacity=getattr(rs, "capacity", None))
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer, alpha)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer, alpha) = ctx.saved_tensors
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, list(rs.tanfovx), list(rs.tanfovy), grad_color, grad_depth,
                grad_alpha, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer,
                alpha, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _C.rasterize_gaussians_backward_batched(*args)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


def rasterize_gaussians_batched(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                raster_settings):
    return _RasterizeGaussiansBatched.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                            cov3Ds_precomp, raster_settings)


def _normalise_optionals(shs, colors_precomp, scales, rotations, cov3D_precomp):
    if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    empty = lambda t: torch.Tensor([]) if t is None else t
    return empty(shs), empty(colors_precomp), empty(scales), empty(rotations), empty(cov3D_precomp)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        raster_settings = self.raster_settings
        shs, colors_precomp, scales, rotations, cov3D_precomp = _normalise_optionals(
            shs, colors_precomp, scales, rotations, cov3D_precomp)
        if isinstance(raster_settings, BatchedRasterizationSettings):
            return rasterize_gaussians_batched(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                               cov3D_precomp, raster_settings)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings)
