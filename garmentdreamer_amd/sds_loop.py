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
rank renders V/N views and ONE flat all-reduce carries every gradient (the scene's own
``grad_bucket``, reduced in place).  The densify / prune schedule of ``on_before_optimizer_step``
(:279-283: every 100 steps in (300, 900), screen-size pruning after step 500) runs with a seeded
generator so that replicas stay identical.  Lightning, OmegaConf and the prompt processor's CLIP
encoder are out of scope.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import dist as gdist
from .cameras import Camera, CameraBatch


class _RenderOut(dict):
    """The dict ``GaussianDreamer.forward`` returns (:212-219); ``opacity = depth / (depth.max() + 1e-5)`` (:215) is
    materialised only if somebody reads it -- the loop's sparsity loss uses the fused head (nn_ops.sparsity_loss)."""

    def __missing__(self, key):
        if key in ("depth_max", "radii_all") and dict.__contains__(self, "_pending_max"):
            # sharded run: the asynchronous [radii | depth maximum] collective started by render_views ends here
            self["radii_all"], self["depth_max"] = dict.pop(self, "_pending_max").finish()
            return self[key]
        if key == "opacity":
            v = self["depth"] / (self["depth_max"] + 1e-5)
            self[key] = v
            return v
        raise KeyError(key)

    # the lazy keys behave like stored ones for every read path of a dict
    def __contains__(self, key):
        if key in ("depth_max", "radii_all") and dict.__contains__(self, "_pending_max"):
            return True
        return key == "opacity" or dict.__contains__(self, key)

    def finish(self):
        """Wait for the pending [radii | depth maximum] collective of a sharded run, if any (a no-op otherwise): after
        this the dict holds plain tensors only and can be copied, iterated or dropped without leaving work in flight."""
        if dict.__contains__(self, "_pending_max"):
            self["depth_max"]
        return self

    def copy(self):
        self.finish()
        return _RenderOut(dict.copy(self))

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def keys(self):
        self.finish()["opacity"]
        return dict.keys(self)

    def items(self):
        self.finish()["opacity"]
        return dict.items(self)

    def values(self):
        self.finish()["opacity"]
        return dict.values(self)

    def __iter__(self):
        self.finish()["opacity"]
        return dict.__iter__(self)


class SDSLoop:
    def __init__(self, gaussians, guidance, prompt_utils, bg_color: torch.Tensor,
                 render_batch_fn: Optional[Callable] = None, lambda_sds: float = 1.0, lambda_sparsity: float = 1.0,
                 lr_scale: float = 1.0, fused_adam: Optional[bool] = None, densify: bool = True,
                 cameras_extent: float = 4.0, densify_seed: int = 0, batch_invariant: bool = False,
                 sync_free: bool = True, capacity_margin: float = 1.5, capacity_quantum: int = 1 << 16, route_as: Optional[tuple] = None):
        self.gaussians = gaussians
        # forward pass without the host read-back of the instance count (rasterizer_impl.cu:282; include/gd_raster.h,
        # gd_raster_forward_batched_capacity): the binning buffer is sized from the previous iterations' counts with a margin
        self.capacity = None
        if sync_free and render_batch_fn is None and gaussians.get_xyz.is_cuda:
            from .diff_gaussian_rasterization._C import InstanceCapacity
            self.capacity = InstanceCapacity(margin=capacity_margin, quantum=capacity_quantum)
        self.batch_invariant = bool(batch_invariant and gaussians.get_xyz.is_cuda)
        # (k, rank) the kernel routing pretends to be; None = the process group's.  `route_as=(8, 0)` on ONE process measures
        # what batch_invariant costs rank 0 of an 8-rank run (bench.py --simulate-world)
        self._route_as = tuple(route_as) if route_as is not None else None
        if self.batch_invariant:
            # sharded run that must reproduce the single-rank gradients as closely as bf16 allows: the guidance
            # kernels are selected for the whole camera batch (views per rank x world size), not for this rank's
            # share (include/gd_nn.h gd_nn_conv_set_route_scale); costs a few % of step time on small shares.
            # Process-global routing state with an OWNER: close() (or dropping the loop) puts it back to 1 only while this
            # loop still owns it (a newer loop's setting survives an older loop's __del__), and step() re-asserts it
            from . import nn_ops
            nn_ops.set_route_scale(*self._route_kr(), owner=self)
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
        self.render_overflow_retries = 0       # iterations whose sync-free render overflowed and was repeated (step())
        self.time_collectives = False          # bench.py --gpus N: time the gradient all-reduce (collective_times())
        self._collective_events = []
        self._bucket = None
        # densify_and_prune(0.0002, 0.05, cameras_extent, size_threshold), GaussianDreamer.py:279-283,426
        self.densify = densify and self.native_scene
        self.cameras_extent = cameras_extent
        self._densify_gen = torch.Generator(device=dev).manual_seed(densify_seed) if self.densify else None
        if not self.native_scene:
            from .gaussian_model import OptimizationParams as a, get_expon_lr_func
            self._xyz_lr = get_expon_lr_func(lr_init=a.position_lr_init * lr_scale,
                                             lr_final=a.position_lr_final * lr_scale,
                                             lr_delay_mult=a.position_lr_delay_mult,
                                             max_steps=a.position_lr_max_steps)

    def _route_kr(self):
        return self._route_as if self._route_as is not None else (gdist.world_size(), gdist.rank())

    def close(self):
        """Undo the process-global kernel routing of ``batch_invariant=True`` (a later loop or guidance in the same process
        would otherwise inherit the k-fold routing)."""
        if getattr(self, "batch_invariant", False):
            from . import nn_ops
            nn_ops.release_route_scale(self)
            self.batch_invariant = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- forward (GaussianDreamer.forward, :180-219) --------------------------------------------
    def render_views(self, batch: Dict, finish: bool = True):
        """``finish=False`` (step() only): in a sharded run the [radii | depth maximum] collective is left in flight and ends
        when "depth_max" / "radii_all" / "opacity" is first read; every other caller gets a finished dict."""
        dev = self.gaussians.get_xyz.device
        c2w = batch["c2w_3dgs"]
        cams = [Camera(c2w[i], batch["fovy"][i], batch["height"], batch["width"], data_device="cpu")
                for i in range(c2w.shape[0])]
        if self.capacity is not None:
            pkg = self.render_batch_fn(CameraBatch(cams, dev), self.gaussians, self.bg, capacity=self.capacity)
        else:
            pkg = self.render_batch_fn(CameraBatch(cams, dev), self.gaussians, self.bg)
        images = pkg["render"].permute(0, 2, 3, 1)      # [V,H,W,3]
        depths = pkg["depth_3dgs"].permute(0, 2, 3, 1)  # [V,H,W,1]
        out = _RenderOut({**pkg, "comp_rgb": images, "depth": depths, "alphas": pkg["alpha"].permute(0, 2, 3, 1)})
        local_max = depths.max()
        if gdist.collectives_on():
            # ONE asynchronous max over the ranks for [radii | depth maximum]; finished in step() after the guidance
            # forward, which it overlaps (dist.PendingMax)
            flag = None
            if self.capacity is not None:
                # the sync-free forward pass leaves {R, R_binned, overflow, capacity} on the device: the overflow flag rides in
                # the collective so that every rank takes the same decision in step() (a synchronising call leaves no flag)
                flag = (self.capacity.buffers(dev)[2:3] if self.capacity.pending
                        else torch.zeros(1, dtype=torch.int32, device=dev))
            out["_pending_max"] = gdist.PendingMax(pkg["radii"].max(dim=0).values, local_max, flag)
            out.flag_source = dict.__getitem__(out, "_pending_max")
            if finish:
                out.finish()
        else:
            out["depth_max"] = local_max
        return out

    def _guidance_may_capture(self, views: int) -> bool:
        """Whether the next guidance call on `views` images may still capture a hipGraph (``guidance.may_capture``; a
        guidance object without that method never captures)."""
        fn = getattr(self.guidance, "may_capture", None)
        return bool(fn(views)) if fn is not None else False

    # -- one iteration ----------------------------------------------------------------------------
    def step(self, batch: Dict, noise=None, timesteps=None, vae_noise=None) -> Dict:
        """``batch``: this rank's shard of the camera batch (keys as uncond.py:395-408)."""
        if hasattr(self.guidance, "update_step"):  # Updateable hook, systems/base.py:148-152
            self.guidance.update_step(0, self.global_step)
        # gaussian.update_learning_rate(true_global_step): exponential xyz schedule, GaussianDreamer.py:231,236
        if self.native_scene:
            self.gaussians.update_learning_rate(self.global_step)
        else:
            for g in self.optimizer.param_groups:
                if g.get("name") == "xyz":
                    g["lr"] = self._xyz_lr(self.global_step)
        if self.global_step > 500:  # GaussianDreamer.py:233-234
            self.guidance.set_min_max_steps(min_step_percent=0.02, max_step_percent=0.55)
        from . import nn_ops
        if self.batch_invariant:
            nn_ops.set_route_scale(*self._route_kr(), owner=self)   # process-global routing: re-asserted
        for attempt in (0, 1):
            out = self.render_views(batch, finish=False)
            if self._guidance_may_capture(out["comp_rgb"].shape[0]):
                # a call of the guidance that may still CAPTURE a hipGraph: no collective of this process is left in flight
                # across a capture (the process group's watchdog thread polls pending work with event queries); once the
                # graphs exist the [radii | depth maximum] collective overlaps their replays
                out.finish()
            g_out = self.guidance(out["comp_rgb"], self.prompt_utils, batch["elevation"], batch["azimuth"],
                                  batch["camera_distances"], rgb_as_latents=False, guidance_eval=False, noise=noise,
                                  timesteps=timesteps, vae_noise=vae_noise)
            # sync-free forward pass: its instance count left for the host right behind the rasterizer's kernels and has
            # landed by now (the guidance forward was queued behind it), so this look costs nothing -- and an overflowed
            # render (nothing binned: every view is the background) must not reach backward / densification / Adam.  The
            # capacity is dropped by overflowed(): the repeat takes the synchronising path (rasterizer_impl.cu:282) and
            # cannot overflow.  With N ranks every rank must take the same decision (the collectives below pair up):
            # the flag is OR-ed over the ranks.
            over = self.capacity is not None and self.capacity.overflowed()
            if self.capacity is not None and getattr(out, "flag_source", None) is not None:
                over = out.flag_source.flag_any()
                out.flag_source = None
            if not over:
                break
            out.finish()
            self.render_overflow_retries += 1
            if attempt == 1:
                raise RuntimeError("rasterizer: the synchronising forward pass reported an overflow")
            self.capacity.value = None
        loss_sds = g_out["loss_sds"]
        radii_all = out.get("radii_all")       # sharded run: max over every rank's views (finishes the pending collective)
        loss_sparsity = nn_ops.sparsity_loss(out["depth"], out["depth_max"])   # mean(sqrt(opacity^2 + 0.01)), :253
        loss = loss_sds * self.lambda_sds + loss_sparsity * self.lambda_sparsity
        if self.native_scene:
            self.gaussians.zero_grad()               # one memset of the flat gradient buffer
        else:
            self.optimizer.zero_grad(set_to_none=True)
        loss.backward()

        densified = False
        with torch.no_grad():
            radii = out["radii"].max(dim=0).values if radii_all is None else radii_all
            if self.native_scene:
                # every parameter's .grad is a view into the scene's bucket; the view-summed viewspace gradient
                # lands in its tail, so ONE in-place all-reduce carries everything (no staging copies)
                vs_grad = self.gaussians.viewspace_grad
                torch.sum(out["viewspace_points"].grad, dim=0, out=vs_grad)
                if gdist.collectives_on():
                    if self.time_collectives and vs_grad.is_cuda:
                        # measured, not assumed (the one data-path collective of the step): events on the step's stream either
                        # side of the flat gradient all-reduce; read with collective_times()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        gdist.all_reduce_mean_(self.gaussians.grad_bucket)
                        e1.record()
                        self._collective_events.append((e0, e1))
                    else:
                        gdist.all_reduce_mean_(self.gaussians.grad_bucket)
                if self.global_step < 900:
                    self.gaussians.add_densification_stats(vs_grad, radii)
                    if self.densify and self.global_step > 300 and self.global_step % 100 == 0:
                        size_threshold = 20 if self.global_step > 500 else None
                        self.gaussians.densify_and_prune(0.0002, 0.05, self.cameras_extent, size_threshold,
                                                         generator=self._densify_gen)
                        densified = True
                        if self.capacity is not None:
                            self.capacity.reset()     # the point count jumped: the next forward pass synchronises once
            else:
                vs_grad = out["viewspace_points"].grad.sum(0)  # sum over this rank's views
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
                for p, g in zip(self.params, grads):
                    p.grad = g
                if gdist.collectives_on():
                    tensors = grads + [vs_grad]
                    if self._bucket is None or not self._bucket.matches(tensors):
                        self._bucket = gdist.GradBucket(tensors)
                    self._bucket.all_reduce_mean_(tensors)
                self._densification_stats(vs_grad, radii)
        if self.native_scene:
            # after densify_and_prune the gradient buffer is fresh (zeros) and the Adam moments of the surviving
            # points were carried over: the optimizer step that follows is the reference's (Lightning calls
            # optimizer.step() right after on_before_optimizer_step; the re-created parameters have no .grad, so
            # Adam skips them, gaussian_model.py:296-340) -- a no-op
            if not densified:
                self.gaussians.step()
        else:
            self.optimizer.step()
        self.global_step += 1
        return {"loss": loss.detach(), "loss_sds": loss_sds.detach(), "loss_sparsity": loss_sparsity.detach(),
                "grad_norm": g_out["grad_norm"], "num_visible": (radii > 0).sum(), "densified": densified}

    def collective_times(self, reset: bool = True):
        """(calls, mean ms, bytes per call) of the flat gradient all-reduce since the last reset (``time_collectives`` on)."""
        ev = self._collective_events
        if not ev:
            return 0, 0.0, 0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        n = len(ev)
        nbytes = self.gaussians.grad_bucket.numel() * self.gaussians.grad_bucket.element_size() if self.native_scene else 0
        if reset:
            self._collective_events = []
        return n, ms, nbytes

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
