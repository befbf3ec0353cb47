"""Scene-side host logic either side of the rasterizer: initialisation from a point
cloud, the six-group Adam with the xyz learning-rate schedule, densification / pruning with optimizer-state
surgery, and the ``.ply`` format -- the host-side mirror of ``GaussianModel``
(Garment_3DGS/gaussiansplatting/scene/gaussian_model.py:123-419), same method names and argument meaning.

MI355X-side differences:
  * ``distCUDA2`` is the HIP kernel ``gd_scene_dist2`` (include/gd_scene.h) -- no CPU path;
  * all parameters live in ONE flat fp32 buffer (+ one flat gradient buffer, + flat Adam moments); the
    ``_xyz ... _rotation`` attributes are views into it, ``.grad`` of each is a view into the flat gradient
    buffer.  One ``gd_scene_adam_step`` launch updates everything (per-group learning rate), one memset clears
    the gradients, and the view-sharded all-reduce sends the flat gradient buffer as it is (no copies);
  * densification re-packs the flat buffers (boolean-mask compaction / concatenation by torch indexing on the
    device) instead of re-creating six ``nn.Parameter`` objects and patching ``optimizer.state``.
The PLY codec is plain numpy (the reference uses ``plyfile``, absent here): binary little-endian
``element vertex`` with float32 properties in the reference's order (``construct_list_of_attributes``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from . import _native
from .scene import SH_C0, GaussianParams, inverse_sigmoid

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")   # order of training_setup's list


def RGB2SH(rgb):
    """utils/sh_utils.py:114-115"""
    return (rgb - 0.5) / SH_C0


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with an optional sine warm-up (utils/general_utils.py:29-62)."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return helper


def build_rotation(r):
    """Unit quaternion (w, x, y, z) rows -> rotation matrices (utils/general_utils.py:78-99)."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest neighbours (simple_knn ``distCUDA2``,
    spatial.cu:14-25) on the GPU.  points: [P,3] float32 CUDA tensor."""
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: the HIP kernel has no CPU path (points must be on the GPU)")
    pts = points.detach().contiguous().float()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    L = _native.lib()
    scratch = torch.empty(L.gd_scene_dist2_scratch_bytes(P), dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        _native.check_scene(L.gd_scene_dist2(torch.cuda.current_stream(pts.device).cuda_stream, P, pts.data_ptr(),
                                             out.data_ptr(), scratch.data_ptr()), "gd_scene_dist2")
    return out


class OptimizationParams:
    """Defaults of this fork's arguments/__init__.py:70-90 (they differ from vanilla 3DGS: slower xyz, 5x faster
    features, 5x slower opacity); pinned by tests/golden/scene_helpers (``optimization_params``)."""
    position_lr_init = 0.00005
    position_lr_final = 0.000025
    position_lr_delay_mult = 0.5
    position_lr_max_steps = 30_000
    feature_lr = 0.0125
    opacity_lr = 0.01
    scaling_lr = 0.005
    rotation_lr = 0.001
    percent_dense = 0.01
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_from_iter = 500
    densify_until_iter = 15_000
    densify_grad_threshold = 0.0002


class _ActivateScene(torch.autograd.Function):
    """``get_features / get_opacity / get_scaling / get_rotation`` (scene/gaussian_model.py:95-115) as one HIP launch
    forward and one backward (gd_scene_activate_*, include/gd_scene.h) instead of exp / sigmoid / normalize / cat
    kernels and their autograd nodes.  With ``grad_views`` (the five slices of the scene's flat gradient buffer)
    the backward accumulates straight into them and returns no gradients: no AccumulateGrad kernels either."""

    @staticmethod
    def forward(ctx, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, grad_views):
        P, M = f_dc.shape[0], 1 + f_rest.shape[1]
        dev = f_dc.device
        shs = torch.empty((P, M, 3), dtype=torch.float32, device=dev)
        opac = torch.empty((P, 1), dtype=torch.float32, device=dev)
        scales = torch.empty((P, 3), dtype=torch.float32, device=dev)
        rots = torch.empty((P, 4), dtype=torch.float32, device=dev)
        L = _native.lib()
        with torch.cuda.device(dev):
            _native.check_scene(L.gd_scene_activate_forward(
                torch.cuda.current_stream(dev).cuda_stream, P, M, f_dc.data_ptr(),
                f_rest.data_ptr() if M > 1 else None, opacity_raw.data_ptr(), scaling_raw.data_ptr(),
                rotation_raw.data_ptr(), shs.data_ptr(), opac.data_ptr(), scales.data_ptr(), rots.data_ptr()),
                "gd_scene_activate_forward")
        ctx.save_for_backward(opac, scales, rotation_raw)
        ctx.grad_views, ctx.M, ctx.rest_shape = grad_views, M, f_rest.shape
        ctx.set_materialize_grads(False)
        return shs, opac, scales, rots

    @staticmethod
    def backward(ctx, d_shs, d_opac, d_scales, d_rots):
        opac, scales, rotation_raw = ctx.saved_tensors
        P, M, dev = opac.shape[0], ctx.M, opac.device
        direct = ctx.grad_views is not None
        if direct:
            g = ctx.grad_views
        else:
            g = (torch.zeros((P, 1, 3), device=dev), torch.zeros(ctx.rest_shape, device=dev),
                 torch.zeros((P, 1), device=dev), torch.zeros((P, 3), device=dev), torch.zeros((P, 4), device=dev))
        c = lambda t: None if t is None else t.contiguous()                       # noqa: E731
        d_shs, d_opac, d_scales, d_rots = c(d_shs), c(d_opac), c(d_scales), c(d_rots)
        p = lambda t: None if t is None else t.data_ptr()                         # noqa: E731
        L = _native.lib()
        with torch.cuda.device(dev):
            _native.check_scene(L.gd_scene_activate_backward(
                torch.cuda.current_stream(dev).cuda_stream, P, M, opac.data_ptr(), scales.data_ptr(),
                rotation_raw.data_ptr(), p(d_shs), p(d_opac), p(d_scales), p(d_rots), g[0].data_ptr(),
                g[1].data_ptr() if M > 1 else None, g[2].data_ptr(), g[3].data_ptr(), g[4].data_ptr()),
                "gd_scene_activate_backward")
        if direct:
            return (None,) * 6
        return g[0], g[1], g[2], g[3], g[4], None


class GaussianModel(GaussianParams):
    """Flat-buffer Gaussian scene with the reference's optimisation / densification surface."""

    def __init__(self, sh_degree: int = 0, device="cuda"):
        nn.Module.__init__(self)
        self.max_sh_degree = sh_degree
        self.active_sh_degree = 0
        self.device = torch.device(device)
        self.spatial_lr_scale = 0.0
        self.percent_dense = 0.0
        self.optimizer_step = 0
        self.lrs: Dict[str, float] = {}
        self._flat = self._grad = self._exp_avg = self._exp_avg_sq = None
        self.generation = -1          # bumped whenever the parameter objects are replaced (_adopt)
        self.xyz_gradient_accum = self.denom = self.max_radii2D = None
        self.xyz_scheduler_args = None

    @classmethod
    def from_activated(cls, scene: dict, sh_degree: int = 0, device="cuda", spatial_lr_scale: float = 1.0):
        """Build from already-activated arrays (``scene.synthetic_gaussians``): scales -> log, opacities ->
        inverse sigmoid, like ``GaussianParams``; then ``training_setup`` with the default rates."""
        m = cls(sh_degree, device)
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=m.device)   # noqa: E731
        shs = t(scene["shs"])
        m._pack({"xyz": t(scene["means3D"]), "f_dc": shs[:, 0:1, :], "f_rest": shs[:, 1:, :],
                 "opacity": inverse_sigmoid(t(scene["opacities"])), "scaling": torch.log(t(scene["scales"])),
                 "rotation": t(scene["rotations"])})
        m.active_sh_degree = sh_degree
        m.spatial_lr_scale = spatial_lr_scale
        m.max_radii2D = torch.zeros((m._xyz.shape[0],), device=m.device)
        m.training_setup()
        return m

    # ---- flat storage -------------------------------------------------------------------------
    def _widths(self) -> List[int]:
        M = (self.max_sh_degree + 1) ** 2
        return [3, 3, 3 * (M - 1), 1, 3, 4]

    def _pack(self, tensors: Dict[str, torch.Tensor], exp_avg=None, exp_avg_sq=None):
        """(Re)build the flat buffers from per-group [P, ...] tensors (+ optional Adam moments)."""
        P = tensors["xyz"].shape[0]
        dev = self.device
        flat = torch.cat([tensors[n].reshape(-1).to(dev, torch.float32) for n in GROUPS])
        m = torch.zeros_like(flat) if exp_avg is None else torch.cat([exp_avg[n].reshape(-1) for n in GROUPS])
        v = torch.zeros_like(flat) if exp_avg_sq is None else torch.cat([exp_avg_sq[n].reshape(-1) for n in GROUPS])
        self._adopt(flat, m, v, P)

    def _adopt(self, flat: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, P: int):
        """Install group-major flat buffers of P points: parameters become views of ``flat``, their ``.grad`` views of
        the (zeroed) gradient bucket.  Every parameter OBJECT is new afterwards: ``generation`` counts these events so
        that a caller holding ``gaussians._xyz`` (or a P-sized buffer) across steps can notice."""
        w = self._widths()
        self._ends = np.cumsum([P * k for k in w]).astype(np.int64)
        dev = self.device
        assert flat.numel() == int(self._ends[-1])
        self._flat = flat
        # [parameter gradients | viewspace gradient summed over views (3 P)]: ONE buffer, so the view-sharded
        # all-reduce (dist.all_reduce_mean_) sends it in place -- no staging copy
        self._bucket = torch.zeros(flat.numel() + 3 * P, dtype=torch.float32, device=dev)
        self._grad = self._bucket[:flat.numel()]
        self.viewspace_grad = self._bucket[flat.numel():].view(P, 3)
        self._exp_avg, self._exp_avg_sq = exp_avg, exp_avg_sq
        self.generation = getattr(self, "generation", -1) + 1
        M = (self.max_sh_degree + 1) ** 2
        shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, M - 1, 3), "opacity": (P, 1), "scaling": (P, 3),
                  "rotation": (P, 4)}
        attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                "scaling": "_scaling", "rotation": "_rotation"}
        start = 0
        for n, end in zip(GROUPS, self._ends):
            p = nn.Parameter(self._flat[start:end].view(shapes[n]))
            p.grad = self._grad[start:end].view(shapes[n])
            setattr(self, attr[n], p)
            start = int(end)

    def _group_views(self, flat) -> Dict[str, torch.Tensor]:
        out, start = {}, 0
        P = self._xyz.shape[0]
        for n, end, w in zip(GROUPS, self._ends, self._widths()):
            out[n] = flat[start:end].view(P, w)
            start = int(end)
        return out

    # ---- initialisation (create_from_pcd, :123-147) -------------------------------------------
    def create_from_pcd(self, points, colors, spatial_lr_scale: float):
        self.spatial_lr_scale = spatial_lr_scale
        dev = self.device
        xyz = torch.as_tensor(np.asarray(points), dtype=torch.float32, device=dev)
        fused_color = RGB2SH(torch.as_tensor(np.asarray(colors), dtype=torch.float32, device=dev))
        P = xyz.shape[0]
        M = (self.max_sh_degree + 1) ** 2
        dist2 = torch.clamp_min(distCUDA2(xyz), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((P, 4), device=dev)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.1 * torch.ones((P, 1), dtype=torch.float, device=dev))
        self._pack({"xyz": xyz, "f_dc": fused_color[:, None, :], "f_rest": torch.zeros((P, M - 1, 3), device=dev),
                    "opacity": opacities, "scaling": scales, "rotation": rots})
        self.max_radii2D = torch.zeros((P,), device=dev)

    # ---- optimisation (training_setup / update_learning_rate, :149-177) -----------------------
    def training_setup(self, training_args=OptimizationParams):
        a = training_args
        self.percent_dense = a.percent_dense
        P = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self.device)
        self.denom = torch.zeros((P, 1), device=self.device)
        if self.max_radii2D is None or self.max_radii2D.shape[0] != P:
            self.max_radii2D = torch.zeros((P,), device=self.device)
        self.lrs = {"xyz": a.position_lr_init * self.spatial_lr_scale, "f_dc": a.feature_lr,
                    "f_rest": a.feature_lr / 20.0, "opacity": a.opacity_lr, "scaling": a.scaling_lr,
                    "rotation": a.rotation_lr}
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=a.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=a.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=a.position_lr_delay_mult,
                                                    max_steps=a.position_lr_max_steps)
        self._exp_avg.zero_()
        self._exp_avg_sq.zero_()
        self.optimizer_step = 0

    def update_learning_rate(self, iteration):
        lr = self.xyz_scheduler_args(iteration)
        self.lrs["xyz"] = lr
        return lr

    def zero_grad(self):
        self._grad.zero_()

    @property
    def grad_bucket(self) -> torch.Tensor:
        """[flat parameter gradient | ``viewspace_grad``] as one contiguous fp32 tensor."""
        return self._bucket

    def oneupSHdegree(self):
        """gaussian_model.py:120-122."""
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def activated(self):
        """(shs [P,M,3], opacity [P,1], scales [P,3], rotations [P,4]) -- the activated accessors in one launch
        (``_ActivateScene``); gradients go straight into the flat gradient buffer.  CPU tensors: the torch ops."""
        if not self._flat.is_cuda:
            return self.get_features, self.get_opacity, self.get_scaling, self.get_rotation
        raw = (self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation)
        views = None
        if torch.is_grad_enabled() and all(p.grad is not None and p.grad.is_contiguous() for p in raw):
            views = tuple(p.grad for p in raw)
        return _ActivateScene.apply(*raw, views)

    def step(self, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15):
        """``torch.optim.Adam(l, lr=0.0, eps=1e-15).step()`` (:166) as one HIP launch over the flat buffer."""
        if not self._flat.is_cuda:
            raise RuntimeError("GaussianModel.step: the HIP Adam kernel has no CPU path")
        self.optimizer_step += 1
        L = _native.lib()
        ends = (C.c_int64 * len(GROUPS))(*[int(e) for e in self._ends])
        lrs = (C.c_double * len(GROUPS))(*[float(self.lrs[n]) for n in GROUPS])
        with torch.cuda.device(self._flat.device):
            _native.check_scene(L.gd_scene_adam_step(
                torch.cuda.current_stream(self._flat.device).cuda_stream, self._flat.data_ptr(), self._grad.data_ptr(),
                self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr(), self._flat.numel(), len(GROUPS), ends, lrs,
                beta1, beta2, eps, self.optimizer_step), "gd_scene_adam_step")

    @property
    def flat_grad(self) -> torch.Tensor:
        """The gradient of every parameter as ONE tensor (what the view-sharded all-reduce sends)."""
        return self._grad

    # ---- densification statistics (add_densification_stats, :415-419; GaussianDreamer.py:268-279) --
    def add_densification_stats(self, viewspace_grad: torch.Tensor, radii: torch.Tensor):
        """viewspace_grad: [P,3] summed over views; radii: [P] int32 max over views."""
        L = _native.lib()
        vg = viewspace_grad.detach().contiguous().float()
        r = radii.contiguous().to(torch.int32)
        with torch.cuda.device(vg.device):
            _native.check_scene(L.gd_scene_densify_stats(
                torch.cuda.current_stream(vg.device).cuda_stream, r.shape[0], r.data_ptr(), vg.data_ptr(),
                self.max_radii2D.data_ptr(), self.xyz_gradient_accum.data_ptr(), self.denom.data_ptr()),
                "gd_scene_densify_stats")

    # ---- densification / pruning (:283-413) ----------------------------------------------------
    def _current(self):
        return ({"xyz": self._xyz.data, "f_dc": self._features_dc.data, "f_rest": self._features_rest.data,
                 "opacity": self._opacity.data, "scaling": self._scaling.data, "rotation": self._rotation.data},
                self._group_views(self._exp_avg), self._group_views(self._exp_avg_sq))

    def prune_points(self, mask: torch.Tensor):
        """Remove the points where ``mask`` is True; Adam moments follow (``_prune_optimizer``)."""
        keep = ~mask
        cur, m, v = self._current()
        self._pack({n: t[keep] for n, t in cur.items()}, {n: t[keep] for n, t in m.items()},
                   {n: t[keep] for n, t in v.items()})
        self.xyz_gradient_accum = self.xyz_gradient_accum[keep]
        self.denom = self.denom[keep]
        self.max_radii2D = self.max_radii2D[keep]

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling,
                              new_rotation):
        """Append new points with zero Adam moments (``cat_tensors_to_optimizer``) and reset the statistics."""
        new = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
               "scaling": new_scaling, "rotation": new_rotation}
        cur, m, v = self._current()
        n_new = new_xyz.shape[0]
        P = cur["xyz"].shape[0]
        z = lambda t: torch.zeros((n_new, t.shape[1]), device=t.device)   # noqa: E731
        self._pack({n: torch.cat((cur[n], new[n].reshape((n_new,) + cur[n].shape[1:])), dim=0) for n in GROUPS},
                   {n: torch.cat((m[n], z(m[n])), dim=0) for n in GROUPS},
                   {n: torch.cat((v[n], z(v[n])), dim=0) for n in GROUPS})
        P = P + n_new
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self.device)
        self.denom = torch.zeros((P, 1), device=self.device)
        self.max_radii2D = torch.zeros((P,), device=self.device)

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        n_init = self.get_xyz.shape[0]
        padded_grad = torch.zeros((n_init,), device=self.device)
        padded_grad[:grads.shape[0]] = grads.squeeze()
        sel = padded_grad >= grad_threshold
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[sel].repeat(N, 1)
        means = torch.zeros((stds.size(0), 3), device=self.device)
        samples = torch.normal(mean=means, std=stds, generator=generator)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = torch.log(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz.detach(), self._features_dc[sel].repeat(N, 1, 1).detach(),
                                   self._features_rest[sel].repeat(N, 1, 1).detach(),
                                   self._opacity[sel].repeat(N, 1).detach(), new_scaling.detach(),
                                   self._rotation[sel].repeat(N, 1).detach())
        prune_filter = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=self.device, dtype=torch.bool)))
        self.prune_points(prune_filter)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel].detach(), self._features_dc[sel].detach(),
                                   self._features_rest[sel].detach(), self._opacity[sel].detach(),
                                   self._scaling[sel].detach(), self._rotation[sel].detach())

    def densify_and_prune_torch(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """The reference's sequence op for op (clone, split, prune through torch indexing; :398-413).  Kept as the
        statement the native path is tested against (tests/test_scene_gpu.py); the training loop calls
        ``densify_and_prune``."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent, generator=generator)
        prune_mask = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            big_points_vs = self.max_radii2D > max_screen_size
            big_points_ws = self.get_scaling.max(dim=1).values > 0.1 * extent
            prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
        self.prune_points(prune_mask)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """``densify_and_prune`` (:398-413) as two HIP passes over the flat buffers (gd_scene_densify_plan / _apply,
        csrc/raster_densify.hip): classification + stream offsets, one host read of the four totals, the seeded normal
        samples of the split (``torch.randn`` = the stream ``torch.normal(mean, std)`` consumes), and one sweep that
        writes parameters and Adam moments of the new point set in the reference's order."""
        if not self._flat.is_cuda:
            raise RuntimeError("GaussianModel.densify_and_prune: the HIP kernels have no CPU path")
        P = self._xyz.shape[0]
        dev = self._flat.device
        L = _native.lib()
        scratch = torch.empty(L.gd_scene_densify_scratch_bytes(P), dtype=torch.uint8, device=dev)
        totals = (C.c_uint32 * 4)()
        dense = float(np.float32(self.percent_dense * extent))
        max_ws = float(np.float32(0.1 * extent)) if max_screen_size else -1.0
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _native.check_scene(L.gd_scene_densify_plan(
                stream, P, self.xyz_gradient_accum.data_ptr(), self.denom.data_ptr(), self._opacity.data_ptr(),
                self._scaling.data_ptr(), float(max_grad), dense, float(min_opacity), max_ws, scratch.data_ptr(), totals),
                "gd_scene_densify_plan", "gd_scene_densify_last_error")
            n_orig, n_clone, n_split, n_child = (int(t) for t in totals)
            newP = n_orig + n_clone + 2 * n_child
            z = None
            if n_split > 0:
                z = torch.randn((2 * n_split, 3), device=dev, dtype=torch.float32, generator=generator)
            w = self._widths()
            per_point = int(sum(w))
            nflat = torch.empty(per_point * newP, dtype=torch.float32, device=dev)
            nm = torch.empty_like(nflat)
            nv = torch.empty_like(nflat)
            widths = (C.c_int * len(w))(*w)
            _native.check_scene(L.gd_scene_densify_apply(
                stream, P, len(w), widths, GROUPS.index("xyz"), GROUPS.index("scaling"), GROUPS.index("rotation"), totals,
                scratch.data_ptr(), z.data_ptr() if z is not None else None, self._flat.data_ptr(),
                self._exp_avg.data_ptr(), self._exp_avg_sq.data_ptr(), nflat.data_ptr(), nm.data_ptr(), nv.data_ptr()),
                "gd_scene_densify_apply", "gd_scene_densify_last_error")
        self._adopt(nflat, nm, nv, newP)
        self.xyz_gradient_accum = torch.zeros((newP, 1), device=dev)
        self.denom = torch.zeros((newP, 1), device=dev)
        self.max_radii2D = torch.zeros((newP,), device=dev)

    def reset_opacity(self):
        """:227-230: clamp opacities to <= 0.01 and zero the opacity group's Adam moments."""
        with torch.no_grad():
            new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
            self._opacity.data.copy_(new)
            lo = int(self._ends[GROUPS.index("opacity") - 1])
            hi = int(self._ends[GROUPS.index("opacity")])
            self._exp_avg[lo:hi].zero_()
            self._exp_avg_sq[lo:hi].zero_()

    # ---- PLY (:187-264) --------------------------------------------------------------------------
    def construct_list_of_attributes(self) -> List[str]:
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path: str):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy()
        f_dc = self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        f_rest = self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        attrs = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, self._opacity.detach().cpu().numpy(),
                                self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy()), axis=1)
        write_ply(path, self.construct_list_of_attributes(), attrs.astype("<f4"))

    def load_ply(self, path: str):
        names, data = read_ply(path)
        col = {n: data[:, i] for i, n in enumerate(names)}
        P = data.shape[0]
        xyz = np.stack((col["x"], col["y"], col["z"]), axis=1)
        f_dc = np.stack((col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]), axis=1)[:, :, None]      # [P,3,1]
        extra = sorted((n for n in names if n.startswith("f_rest_")), key=lambda n: int(n.split("_")[-1]))
        assert len(extra) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        f_extra = np.stack([col[n] for n in extra], axis=1).reshape(P, 3, (self.max_sh_degree + 1) ** 2 - 1) \
            if extra else np.zeros((P, 3, 0), np.float32)
        scales = np.stack([col[n] for n in sorted((n for n in names if n.startswith("scale_")),
                                                  key=lambda n: int(n.split("_")[-1]))], axis=1)
        rots = np.stack([col[n] for n in sorted((n for n in names if n.startswith("rot")),
                                                key=lambda n: int(n.split("_")[-1]))], axis=1)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float, device=self.device)   # noqa: E731
        self._pack({"xyz": t(xyz), "f_dc": t(f_dc).transpose(1, 2).contiguous(),
                    "f_rest": t(f_extra).transpose(1, 2).contiguous(), "opacity": t(col["opacity"][:, None]),
                    "scaling": t(scales), "rotation": t(rots)})
        self.active_sh_degree = self.max_sh_degree
        self.max_radii2D = torch.zeros((P,), device=self.device)


# ---------------------------------------------------------------------------------------------------
# minimal PLY codec for the ``last_3dgs.ply`` layout (one ``vertex`` element, float32 scalar properties)
# ---------------------------------------------------------------------------------------------------

def write_ply(path: str, names: List[str], data: np.ndarray):
    """binary_little_endian 1.0, ``element vertex N``, ``property float <name>`` per column -- the bytes
    ``plyfile``'s ``PlyData([PlyElement.describe(elements, 'vertex')]).write(path)`` produces on a little-endian
    host for an all-'f4' structured array (gaussian_model.py:199-206)."""
    assert data.ndim == 2 and data.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {data.shape[0]}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(data, dtype="<f4").tobytes())


def read_ply(path: str):
    """Reads what ``write_ply`` / plyfile write (binary little/big endian or ascii; float/double/int scalar
    properties of the first element).  Returns (property names, float32 array [N, len(names)])."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, names, types, count, in_first, seen = None, [], [], 0, False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header without end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError("list properties are not part of the Gaussian PLY layout")
                types.append(tok[1])
                names.append(tok[2])
            elif tok[0] == "end_header":
                break
        np_t = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "int": "i4", "int32": "i4",
                "uint": "u4", "uint32": "u4", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
                "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1"}
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(count)]
            data = np.array(rows, dtype=np.float64).reshape(count, len(names)).astype(np.float32)
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, e + np_t[t]) for n, t in zip(names, types)])
            raw = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
            data = np.stack([raw[n].astype(np.float32) for n in names], axis=1) if names else np.zeros((count, 0), np.float32)
    return names, data
