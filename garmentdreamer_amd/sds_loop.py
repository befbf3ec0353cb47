"""The per-iteration inner loop of GarmentDreamer stage 1, as plain Python over the MI355X ops.

Iteration structure restated from ``GaussianDreamer.forward / training_step /
on_before_optimizer_step`` (Garment_3DGS/threestudio/systems/GaussianDreamer.py:180-283):

    render V views -> comp_rgb [V,H,W,3], depth [V,H,W,1]
    opacity  = depth / (depth.max() + 1e-5)                               (:215)
    loss     = lambda_sds * guidance(comp_rgb, ...)["loss_sds"]
             + lambda_sparsity * mean(sqrt(opacity^2 + 0.01))            (:248-255)
    backward; sum_v viewspace.grad, max_v radii -> densification stats   (:268-279)
    Adam(eps=1e-15) over the six parameter groups                         (gaussian_model.py:156-167)

Differences, all MI355X-side: the V views go through ONE batched rasterizer launch set
(``render_batch``) instead of a Python loop of V launch sets + V host syncs; with N ranks each
rank renders V/N views and ONE flat all-reduce carries every gradient (``dist.GradBucket``).
Lightning, OmegaConf, the prompt processor's CLIP encoder, densify/prune and PLY export are out
of scope.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import dist as gdist
from .cameras import Camera, CameraBatch


class SDSLoop:
    def __init__(self, gaussians, guidance, prompt_utils, bg_color: torch.Tensor,
                 render_batch_fn: Optional[Callable] = None, lambda_sds: float = 1.0, lambda_sparsity: float = 1.0,
                 lr_scale: float = 1.0, fused_adam: Optional[bool] = None):
        self.gaussians = gaussians
        self.guidance = guidance
        self.prompt_utils = prompt_utils
        self.bg = bg_color
        if render_batch_fn is None:
            from .gaussian_renderer import render_batch as render_batch_fn  # HIP rasterizer (no fallback)
        self.render_batch_fn = render_batch_fn
        self.lambda_sds, self.lambda_sparsity = lambda_sds, lambda_sparsity
        dev = gaussians.get_xyz.device
        # gaussian_model.GaussianModel: flat parameter / gradient buffers, one HIP Adam launch, native statistics
        self.native_scene = hasattr(gaussians, "flat_grad")
        if self.native_scene:
            self.optimizer = None
            self.params = []
        else:
            groups = gaussians.param_groups()
            for g in groups:
                g["lr"] *= lr_scale
            if fused_adam is None:
                fused_adam = dev.type == "cuda"
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=fused_adam)
            self.params = [p for g in groups for p in g["params"]]
            P = gaussians.get_xyz.shape[0]
            self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
            self.denom = torch.zeros((P, 1), device=dev)
            self.max_radii2D = torch.zeros((P,), device=dev)
        self.global_step = 0
        self._bucket = None

    # -- forward (GaussianDreamer.forward, :180-219) --------------------------------------------
    def render_views(self, batch: Dict):
        dev = self.gaussians.get_xyz.device
        c2w = batch["c2w_3dgs"]
        cams = [Camera(c2w[i], batch["fovy"][i], batch["height"], batch["width"], data_device="cpu")
                for i in range(c2w.shape[0])]
        pkg = self.render_batch_fn(CameraBatch(cams, dev), self.gaussians, self.bg)
        images = pkg["render"].permute(0, 2, 3, 1)      # [V,H,W,3]
        depths = pkg["depth_3dgs"].permute(0, 2, 3, 1)  # [V,H,W,1]
        dmax = gdist.global_max(depths.max())
        return {**pkg, "comp_rgb": images, "depth": depths, "opacity": depths / (dmax + 1e-5),
                "alphas": pkg["alpha"].permute(0, 2, 3, 1)}

    # -- one iteration ----------------------------------------------------------------------------
    def step(self, batch: Dict, noise=None, timesteps=None, vae_noise=None) -> Dict:
        """``batch``: this rank's shard of the camera batch (keys as uncond.py:395-408)."""
        if hasattr(self.guidance, "update_step"):  # Updateable hook, systems/base.py:148-152
            self.guidance.update_step(0, self.global_step)
        if self.global_step > 500:  # GaussianDreamer.py:233-234
            self.guidance.set_min_max_steps(min_step_percent=0.02, max_step_percent=0.55)
        out = self.render_views(batch)
        g_out = self.guidance(out["comp_rgb"], self.prompt_utils, batch["elevation"], batch["azimuth"],
                              batch["camera_distances"], rgb_as_latents=False, guidance_eval=False, noise=noise,
                              timesteps=timesteps, vae_noise=vae_noise)
        loss_sds = g_out["loss_sds"]
        loss_sparsity = (out["opacity"] ** 2 + 0.01).sqrt().mean()
        loss = loss_sds * self.lambda_sds + loss_sparsity * self.lambda_sparsity
        if self.native_scene:
            self.gaussians.zero_grad()               # one memset of the flat gradient buffer
        else:
            self.optimizer.zero_grad(set_to_none=True)
        loss.backward()

        with torch.no_grad():
            vs_grad = out["viewspace_points"].grad.sum(0)  # sum over this rank's views
            radii = out["radii"].max(dim=0).values
            if self.native_scene:
                grads = [self.gaussians.flat_grad]   # every parameter's .grad is a view into this buffer
            else:
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
                for p, g in zip(self.params, grads):
                    p.grad = g
            if gdist.world_size() > 1:
                tensors = grads + [vs_grad]
                if self._bucket is None:
                    self._bucket = gdist.GradBucket(tensors)
                self._bucket.all_reduce_mean_(tensors)
                gdist.all_reduce_max_(radii)
            if self.native_scene:
                if self.global_step < 900:
                    self.gaussians.add_densification_stats(vs_grad, radii)
            else:
                self._densification_stats(vs_grad, radii)
        if self.native_scene:
            self.gaussians.step()
        else:
            self.optimizer.step()
        self.global_step += 1
        return {"loss": loss.detach(), "loss_sds": loss_sds.detach(), "loss_sparsity": loss_sparsity.detach(),
                "grad_norm": g_out["grad_norm"], "num_visible": (radii > 0).sum()}

    def _densification_stats(self, viewspace_grad, radii):
        """on_before_optimizer_step (:268-279) + add_densification_stats (gaussian_model.py:415-419)."""
        if self.global_step >= 900:
            return
        vis = radii > 0
        self.max_radii2D = torch.where(vis, torch.max(self.max_radii2D, radii.to(self.max_radii2D.dtype)),
                                       self.max_radii2D)
        self.xyz_gradient_accum += torch.where(vis[:, None], viewspace_grad[:, :2].norm(dim=-1, keepdim=True),
                                               torch.zeros_like(self.xyz_gradient_accum))
        self.denom += vis[:, None].to(self.denom.dtype)
