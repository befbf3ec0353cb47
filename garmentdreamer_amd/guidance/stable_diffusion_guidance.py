"""Score-Distillation guidance with the threestudio plugin surface.

Mirrors ``StableDiffusionGuidance`` (Garment_3DGS/threestudio/models/guidance/
stable_diffusion_guidance.py): same ``Config`` fields (:21-48), ``__call__`` signature and output
dict (:374-448), ``set_min_max_steps`` (:141-143), ``update_step`` (:581-591), ``encode_images``
(:160-167), ``forward_unet`` (:146-157), ``compute_grad_sds`` (:185-276) -- including the
reference's CFG form ``eps_text + s (eps_text - eps_uncond)`` (:249-251) with cond-first text
embeddings (prompt_processors/base.py:77-78), ``w = 1 - alphas_cumprod[t]``, nan_to_num, the
``grad_clip`` schedule and the reparameterised MSE loss whose gradient w.r.t. the latents is
exactly ``grad`` (:418-427).

What differs, on purpose:
  * the UNet / VAE / scheduler come from ``sd21.py`` (own restatement, random-init unless a local
    checkpoint path is given) instead of ``diffusers.StableDiffusionPipeline.from_pretrained``;
  * half precision is bf16 on MI355X (reference: fp16 + Lightning GradScaler);
  * randomness is injectable: ``__call__(..., noise=, timesteps=, vae_noise=)`` so tests and
    multi-GPU runs are reproducible (the reference draws VAE noise -> t -> eps from the global
    generator, in that order: :166,401-407,237).  Without them the same order is used.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _runtime_env, nn_ops
from . import sd21


def C(value: Any, epoch: int, global_step: int) -> float:
    """threestudio's schedule literal ``[start_step, v0, v1, end_step]``
    (Garment_3DGS/threestudio/utils/misc.py:65-86)."""
    if isinstance(value, (int, float)):
        return value
    value = list(value)
    if len(value) == 3:
        value = [0] + value
    assert len(value) == 4
    start_step, start_value, end_value, end_step = value
    current = global_step if isinstance(end_step, int) else epoch
    return start_value + (end_value - start_value) * max(min(1.0, (current - start_step) / (end_step - start_step)), 0.0)


class StableDiffusionGuidance(nn.Module):
    @dataclass
    class Config:
        pretrained_model_name_or_path: str = "stabilityai/stable-diffusion-2-1-base"
        enable_memory_efficient_attention: bool = False
        enable_sequential_cpu_offload: bool = False
        enable_attention_slicing: bool = False
        enable_channels_last_format: bool = True
        guidance_scale: float = 100.0
        grad_clip: Optional[Any] = None
        half_precision_weights: bool = True
        min_step_percent: float = 0.02
        max_step_percent: float = 0.98
        max_step_percent_annealed: float = 0.5
        anneal_start_step: Optional[int] = None
        use_sjc: bool = False
        var_red: bool = True
        weighting_strategy: str = "sds"
        token_merging: bool = False
        token_merging_params: Optional[dict] = field(default_factory=dict)
        view_dependent_prompting: bool = True
        max_items_eval: int = 4
        # additions
        unet_weights: Optional[str] = None   # local diffusers safetensors; None -> random init
        vae_weights: Optional[str] = None
        init_seed: int = 0
        # Replay the UNet forward and the VAE encoder forward/backward as hipGraphs (one graph per
        # batch shape).  The step is launch-bound once a GPU holds only 1-2 views (8-GPU sharding:
        # ~1500 kernels of ~5 us each); graphs remove the per-kernel host cost.  Numerics unchanged.
        use_hip_graphs: bool = False
        # e4m3 3x3 convolutions in the no-grad UNet forward (BASELINE configs[4], nn_ops.Fp8State): the first
        # forward_unet call calibrates the activation ranges in bf16, later calls run fp8.  bf16 stays the default.
        fp8_unet: bool = False
        # eager bf16 forwards that record the activation ranges before the e4m3 path (and the graph capture) starts:
        # each draws its own timesteps, so several of them span the t range the static scales have to cover
        fp8_calibration_steps: int = 3

    def __init__(self, cfg: Optional[dict] = None, device="cuda", unet: Optional[nn.Module] = None,
                 vae: Optional[nn.Module] = None):
        super().__init__()
        cfg = cfg or {}
        self.cfg = cfg if isinstance(cfg, self.Config) else self.Config(**cfg)
        self.device = torch.device(device)
        self.configure(unet, vae)

    def configure(self, unet=None, vae=None) -> None:
        self.weights_dtype = torch.bfloat16 if self.cfg.half_precision_weights else torch.float32
        if unet is None:
            with torch.device(self.device):
                unet = sd21.UNet2DConditionModel()
            if self.cfg.unet_weights:
                sd21.load_diffusers_weights(unet, self.cfg.unet_weights)
            else:
                sd21.init_random_(unet, self.cfg.init_seed)
        if vae is None:
            with torch.device(self.device):
                vae = sd21.AutoencoderKLEncoder()
            if self.cfg.vae_weights:
                sd21.load_diffusers_weights(vae, self.cfg.vae_weights)
            else:
                sd21.init_random_(vae, self.cfg.init_seed + 1)
        fmt = torch.channels_last if self.cfg.enable_channels_last_format else torch.contiguous_format
        self.unet = unet.to(device=self.device, dtype=self.weights_dtype).to(memory_format=fmt).eval()
        self.vae = vae.to(device=self.device, dtype=self.weights_dtype).to(memory_format=fmt).eval()
        for p in self.vae.parameters():
            p.requires_grad_(False)
        for p in self.unet.parameters():
            p.requires_grad_(False)
        self.scheduler = sd21.DDIMScheduler()
        self.num_train_timesteps = self.scheduler.config.num_train_timesteps
        self.set_min_max_steps()
        self.alphas = self.scheduler.alphas_cumprod.to(self.device)
        if self.cfg.use_sjc:
            # score-jacobian chaining (:109-118, :132-134; off in GarmentDreamer's config): the reference swaps in a
            # DDPMScheduler with the SAME scaled-linear betas 0.00085 .. 0.012, i.e. the same alphas_cumprod; only
            # sigma_t = sqrt((1 - abar_t) / abar_t) is new
            self.us = torch.sqrt((1 - self.alphas) / self.alphas)
        self.grad_clip_val: Optional[float] = None
        self._unet_graphs = {}
        self._vae_graphs = {}
        self._fp8_calibrated = False
        self._fp8_calib_done = 0
        if self.cfg.fp8_unet and self.weights_dtype == torch.bfloat16 and self.device.type == "cuda":
            self.unet.enable_fp8()
        if self.cfg.use_hip_graphs and not _runtime_env.graph_replay_safe():
            import warnings
            warnings.warn(f"{_runtime_env.FLAG}=0 was not in place before the HIP runtime started; hipGraph replay of "
                          "the UNet / VAE is off for this process (eager launches). Import garmentdreamer_amd before "
                          "the first torch.cuda call, or export the flag.")
            self.cfg.use_hip_graphs = False

    def set_min_max_steps(self, min_step_percent=0.02, max_step_percent=0.98):
        self.min_step = int(self.num_train_timesteps * min_step_percent)
        self.max_step = int(self.num_train_timesteps * max_step_percent)

    def forward_unet(self, latents, t, encoder_hidden_states):
        input_dtype = latents.dtype
        # the reference casts t to its fp16 weights dtype (:155), which holds every timestep < 2048 exactly; bf16
        # would round t > 256 to a multiple of 2 or 4, so with bf16 weights the timestep stays fp32
        t_dtype = torch.float32 if self.weights_dtype == torch.bfloat16 else self.weights_dtype
        x, tt, ctx = latents.to(self.weights_dtype), t.to(t_dtype), encoder_hidden_states.to(self.weights_dtype)
        fp8 = getattr(self.unet, "fp8", None)
        if fp8 is not None and not self._fp8_calibrated and not torch.is_grad_enabled():
            # one eager bf16 forward on the real inputs records every fp8 site's activation range
            out = self.unet(x, tt, encoder_hidden_states=ctx).to(input_dtype)
            self._fp8_calib_done += 1
            if self._fp8_calib_done >= max(1, int(self.cfg.fp8_calibration_steps)):
                fp8.mode = "run"
                self._fp8_calibrated = True
            return out
        if self.cfg.use_hip_graphs and x.is_cuda and not torch.is_grad_enabled():
            try:
                return self._graphed_unet(x, tt, ctx).to(input_dtype)
            except RuntimeError as e:      # capture refused (driver / allocator state): keep running, eagerly
                self._graphs_failed(e)
        return self.unet(x, tt, encoder_hidden_states=ctx).to(input_dtype)

    def may_capture(self, batch_size: int) -> bool:
        """True while a call on `batch_size` images may still CAPTURE a hipGraph (the caller then keeps no collective of its
        process in flight across it): graphs are on and the UNet / VAE graph of that batch shape does not exist yet."""
        if not self.cfg.use_hip_graphs:
            return False
        have_u = any(k[0][0] in (2 * batch_size, 4 * batch_size) for k in self._unet_graphs)
        have_v = any(k[0] == batch_size for k in self._vae_graphs)
        return not (have_u and have_v)

    def _graphs_failed(self, err):
        """hipGraph capture is an optimisation: if it cannot be set up, say so once and continue with eager launches."""
        import warnings
        warnings.warn(f"hipGraph capture failed ({err}); continuing with eager kernel launches")
        self.cfg.use_hip_graphs = False
        self._unet_graphs.clear()
        self._vae_graphs.clear()
        torch.cuda.synchronize()
        from .. import nn_ops
        nn_ops.reset_workspaces()      # an aborted capture may have left a statistics workspace mid-update

    # ---- hipGraph replay ----------------------------------------------------------------------
    def _graphed_unet(self, x, t, ctx):
        key = (tuple(x.shape), tuple(ctx.shape))
        entry = self._unet_graphs.get(key)
        if entry is None:
            sx, st, sc = x.clone(), t.clone(), ctx.clone()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):   # warm-up outside capture: library solver selection, weight caches
                    self.unet(sx, st, encoder_hidden_states=sc)
            torch.cuda.current_stream(x.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                out = self.unet(sx, st, encoder_hidden_states=sc)
            entry = self._unet_graphs[key] = (g, sx, st, sc, out)
        g, sx, st, sc, out = entry
        sx.copy_(x)
        st.copy_(t)
        sc.copy_(ctx)
        g.replay()
        return out.clone()

    def _graphed_vae_moments(self, imgs):
        """``quant_conv(encoder(imgs))`` with forward AND backward replayed from hipGraphs
        (torch.cuda.make_graphed_callables builds the autograd-aware pair)."""
        key = tuple(imgs.shape)
        fn = self._vae_graphs.get(key)
        if fn is None:
            vae = self.vae

            class _Moments(nn.Module):
                def forward(self, x):
                    return vae.quant_conv(vae.encoder(x))

            sample = torch.rand(imgs.shape, device=imgs.device, dtype=imgs.dtype, requires_grad=True)
            fn = self._vae_graphs[key] = torch.cuda.make_graphed_callables(_Moments(), (sample,))
        return fn(imgs)

    def encode_images(self, imgs, vae_noise: Optional[torch.Tensor] = None):
        return self._encode_prepared((imgs * 2.0 - 1.0).to(self.weights_dtype), vae_noise, imgs.dtype)

    def _encode_prepared(self, x, vae_noise, out_dtype):
        """``x``: images already mapped to [-1, 1] in the VAE's dtype (``encode_images`` or the fused prologue)."""
        with nn_ops.route_batch(1, x.shape[0]):
            if self.cfg.use_hip_graphs and x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
                xc = x.contiguous(memory_format=torch.channels_last)
                try:
                    posterior = sd21.DiagonalGaussianDistribution(self._graphed_vae_moments(xc))
                except RuntimeError as e:
                    self._graphs_failed(e)
                    posterior = self.vae.encode(x).latent_dist
            else:
                posterior = self.vae.encode(x).latent_dist
        noise = None if vae_noise is None else vae_noise.to(self.weights_dtype)
        latents = posterior.sample(noise) * self.vae.config.scaling_factor
        return latents.to(out_dtype)

    def compute_grad_sds(self, latents, t, prompt_utils, elevation, azimuth, camera_distances,
                         noise: Optional[torch.Tensor] = None):
        use_perp_neg = bool(getattr(prompt_utils, "use_perp_neg", False))
        batch_size = elevation.shape[0]
        neg_guidance_weights = None
        if use_perp_neg:
            # perp-neg prompting (:196-228; off in GarmentDreamer's config): one UNet call on 4B samples --
            # [interpolated positive | uncond | two negative prompts per view] -- and the negative directions enter only
            # through their component perpendicular to the positive one (threestudio/utils/ops.py:431-441).  Note the
            # plain classifier-free form here, uncond + s (e_pos + ...), unlike the text + s (text - uncond) below.
            text_embeddings, neg_guidance_weights = prompt_utils.get_text_embeddings_perp_neg(
                elevation, azimuth, camera_distances, self.cfg.view_dependent_prompting)
            with torch.no_grad():
                if noise is None:
                    noise = torch.randn_like(latents)
                latents_noisy = self.scheduler.add_noise(latents, noise, t)
                with nn_ops.route_batch(4, 4 * batch_size):
                    noise_pred = self.forward_unet(torch.cat([latents_noisy] * 4, dim=0), torch.cat([t] * 4),
                                                   encoder_hidden_states=text_embeddings)
            noise_pred_text = noise_pred[:batch_size]
            noise_pred_uncond = noise_pred[batch_size:batch_size * 2]
            noise_pred_neg = noise_pred[batch_size * 2:]
            e_pos = noise_pred_text - noise_pred_uncond
            accum_grad = 0
            n_negative_prompts = neg_guidance_weights.shape[-1]
            for i in range(n_negative_prompts):
                e_i_neg = noise_pred_neg[i::n_negative_prompts] - noise_pred_uncond
                accum_grad = accum_grad + neg_guidance_weights[:, i].view(-1, 1, 1, 1).to(e_pos) * \
                    perpendicular_component(e_i_neg, e_pos)
            noise_pred = noise_pred_uncond + self.cfg.guidance_scale * (e_pos + accum_grad)
        else:
            text_embeddings = prompt_utils.get_text_embeddings(elevation, azimuth, camera_distances,
                                                               self.cfg.view_dependent_prompting)
            with torch.no_grad():
                if noise is None:
                    noise = torch.randn_like(latents)
                latents_noisy = self.scheduler.add_noise(latents, noise, t)
                latent_model_input = torch.cat([latents_noisy] * 2, dim=0)
                with nn_ops.route_batch(2, 2 * batch_size):
                    noise_pred = self.forward_unet(latent_model_input, torch.cat([t] * 2),
                                                   encoder_hidden_states=text_embeddings)
            noise_pred_text, noise_pred_uncond = noise_pred.chunk(2)
            noise_pred = noise_pred_text + self.cfg.guidance_scale * (noise_pred_text - noise_pred_uncond)

        if self.cfg.weighting_strategy == "sds":
            w = (1 - self.alphas[t]).view(-1, 1, 1, 1)
        elif self.cfg.weighting_strategy == "uniform":
            w = 1
        elif self.cfg.weighting_strategy == "fantasia3d":
            w = (self.alphas[t] ** 0.5 * (1 - self.alphas[t])).view(-1, 1, 1, 1)
        else:
            raise ValueError(f"Unknown weighting strategy: {self.cfg.weighting_strategy}")
        grad = w * (noise_pred - noise)
        guidance_eval_utils = {"use_perp_neg": use_perp_neg, "neg_guidance_weights": neg_guidance_weights,
                               "text_embeddings": text_embeddings, "t_orig": t, "latents_noisy": latents_noisy,
                               "noise_pred": noise_pred}
        return grad, guidance_eval_utils

    def compute_grad_sjc(self, latents, t, prompt_utils, elevation, azimuth, camera_distances,
                         noise: Optional[torch.Tensor] = None):
        """Score-jacobian-chaining form of the latent gradient (reference :278-372; ``use_sjc``, off in
        GarmentDreamer's config): the latent is perturbed in the variance-EXPLODING parametrisation
        ``z = y + sigma_t n``, the UNet sees ``z / sqrt(1 + sigma_t^2)``, and with the denoised estimate
        ``D = z - sigma_t eps`` the gradient is ``-(D - y) / sigma_t`` (``var_red``) or ``-(D - z) / sigma_t``."""
        use_perp_neg = bool(getattr(prompt_utils, "use_perp_neg", False))
        batch_size = elevation.shape[0]
        sigma = self.us[t].view(-1, 1, 1, 1)
        neg_guidance_weights = None
        if use_perp_neg:
            text_embeddings, neg_guidance_weights = prompt_utils.get_text_embeddings_perp_neg(
                elevation, azimuth, camera_distances, self.cfg.view_dependent_prompting)
            reps = 4
        else:
            text_embeddings = prompt_utils.get_text_embeddings(elevation, azimuth, camera_distances,
                                                               self.cfg.view_dependent_prompting)
            reps = 2
        with torch.no_grad():
            if noise is None:
                noise = torch.randn_like(latents)
            y = latents
            zs = y + sigma * noise
            scaled_zs = zs / torch.sqrt(1 + sigma ** 2)
            with nn_ops.route_batch(reps, reps * batch_size):
                noise_pred = self.forward_unet(torch.cat([scaled_zs] * reps, dim=0), torch.cat([t] * reps),
                                               encoder_hidden_states=text_embeddings)
        if use_perp_neg:
            noise_pred_text = noise_pred[:batch_size]
            noise_pred_uncond = noise_pred[batch_size:batch_size * 2]
            noise_pred_neg = noise_pred[batch_size * 2:]
            e_pos = noise_pred_text - noise_pred_uncond
            accum_grad = 0
            n_negative_prompts = neg_guidance_weights.shape[-1]
            for i in range(n_negative_prompts):
                e_i_neg = noise_pred_neg[i::n_negative_prompts] - noise_pred_uncond
                accum_grad = accum_grad + neg_guidance_weights[:, i].view(-1, 1, 1, 1).to(e_pos) * \
                    perpendicular_component(e_i_neg, e_pos)
            noise_pred = noise_pred_uncond + self.cfg.guidance_scale * (e_pos + accum_grad)
        else:
            noise_pred_text, noise_pred_uncond = noise_pred.chunk(2)
            noise_pred = noise_pred_text + self.cfg.guidance_scale * (noise_pred_text - noise_pred_uncond)
        Ds = zs - sigma * noise_pred
        grad = -(Ds - y) / sigma if self.cfg.var_red else -(Ds - zs) / sigma
        guidance_eval_utils = {"use_perp_neg": use_perp_neg, "neg_guidance_weights": neg_guidance_weights,
                               "text_embeddings": text_embeddings, "t_orig": t, "latents_noisy": scaled_zs,
                               "noise_pred": noise_pred}
        return grad, guidance_eval_utils

    def __call__(self, rgb, prompt_utils, elevation, azimuth, camera_distances, rgb_as_latents=False,
                 guidance_eval=False, noise: Optional[torch.Tensor] = None, timesteps: Optional[torch.Tensor] = None,
                 vae_noise: Optional[torch.Tensor] = None, **kwargs):
        batch_size = rgb.shape[0]
        rgb_BCHW = rgb.permute(0, 3, 1, 2)
        if rgb_as_latents:
            latents = F.interpolate(rgb_BCHW, (64, 64), mode="bilinear", align_corners=False)
        elif self.weights_dtype == torch.bfloat16 and nn_ops.vae_prologue_supported(rgb_BCHW):
            # bilinear 512^2 + (2x - 1) + bf16 / NHWC cast: ONE launch each way (csrc/nn_prologue.hip) instead of
            # interpolate, mul-add, dtype copy, layout copy and their backward kernels
            latents = self._encode_prepared(nn_ops.vae_prologue(rgb_BCHW, 512, 512), vae_noise, rgb.dtype)
        else:
            rgb_BCHW_512 = F.interpolate(rgb_BCHW, (512, 512), mode="bilinear", align_corners=False)
            latents = self.encode_images(rgb_BCHW_512, vae_noise)

        if timesteps is None:
            t = torch.randint(self.min_step, self.max_step + 1, [batch_size], dtype=torch.long, device=self.device)
        else:
            t = timesteps.to(device=self.device, dtype=torch.long)

        compute_grad = self.compute_grad_sjc if self.cfg.use_sjc else self.compute_grad_sds
        grad, guidance_eval_utils = compute_grad(latents, t, prompt_utils, elevation, azimuth, camera_distances, noise)
        grad = torch.nan_to_num(grad)
        if self.grad_clip_val is not None:
            grad = grad.clamp(-self.grad_clip_val, self.grad_clip_val)
        target = (latents - grad).detach()
        loss_sds = 0.5 * F.mse_loss(latents, target, reduction="sum") / batch_size
        guidance_out = {"loss_sds": loss_sds, "grad_norm": grad.norm(), "min_step": self.min_step,
                        "max_step": self.max_step}
        if guidance_eval:
            # debugging previews (:436-446; GaussianDreamer.py:244 passes guidance_eval=False)
            guidance_eval_out = self.guidance_eval(**guidance_eval_utils, generator=kwargs.get("eval_generator"))
            texts = []
            for n, e, a, c in zip(guidance_eval_out["noise_levels"], elevation, azimuth, camera_distances):
                texts.append(f"n{n:.02f}\ne{e.item():.01f}\na{a.item():.01f}\nc{c.item():.02f}")
            guidance_eval_out.update({"texts": texts})
            guidance_out.update({"eval": guidance_eval_out})
        return guidance_out

    def decode_latents(self, latents, latent_height: int = 64, latent_width: int = 64):
        """latents -> [B, 3, 8 h, 8 w] image in [0, 1] (:170-183).  The VAE DECODER is not part of the per-iteration path and
        not restated in sd21: a VAE object that has ``decode`` (e.g. diffusers' AutoencoderKL) must be supplied."""
        if not hasattr(self.vae, "decode"):
            raise RuntimeError("guidance_eval needs a VAE with a decoder (pass vae=<object with .decode>); the restated "
                               "AutoencoderKLEncoder has the encoder half only")
        input_dtype = latents.dtype
        latents = F.interpolate(latents, (latent_height, latent_width), mode="bilinear", align_corners=False)
        latents = 1 / self.vae.config.scaling_factor * latents
        image = self.vae.decode(latents.to(self.weights_dtype)).sample
        image = (image * 0.5 + 0.5).clamp(0, 1)
        return image.to(input_dtype)

    @torch.no_grad()
    def get_noise_pred(self, latents_noisy, t, text_embeddings, use_perp_neg=False, neg_guidance_weights=None):
        """Guided noise prediction of ONE timestep for the preview sampler (:452-501)."""
        batch_size = latents_noisy.shape[0]
        reps = 4 if use_perp_neg else 2
        from .. import nn_ops
        with nn_ops.route_batch(reps, reps * batch_size):     # (read under batch-invariant kernel selection only)
            noise_pred = self.forward_unet(torch.cat([latents_noisy] * reps, dim=0),
                                           torch.cat([t.reshape(1)] * reps).to(self.device),
                                           encoder_hidden_states=text_embeddings)
        if use_perp_neg:
            noise_pred_text = noise_pred[:batch_size]
            noise_pred_uncond = noise_pred[batch_size:batch_size * 2]
            noise_pred_neg = noise_pred[batch_size * 2:]
            e_pos = noise_pred_text - noise_pred_uncond
            accum_grad = 0
            n_negative_prompts = neg_guidance_weights.shape[-1]
            for i in range(n_negative_prompts):
                e_i_neg = noise_pred_neg[i::n_negative_prompts] - noise_pred_uncond
                accum_grad = accum_grad + neg_guidance_weights[:, i].view(-1, 1, 1, 1).to(e_pos) * \
                    perpendicular_component(e_i_neg, e_pos)
            return noise_pred_uncond + self.cfg.guidance_scale * (e_pos + accum_grad)
        noise_pred_text, noise_pred_uncond = noise_pred.chunk(2)
        return noise_pred_text + self.cfg.guidance_scale * (noise_pred_text - noise_pred_uncond)

    @torch.no_grad()
    def guidance_eval(self, t_orig, text_embeddings, latents_noisy, noise_pred, use_perp_neg=False,
                      neg_guidance_weights=None, generator=None):
        """Previews of what the guidance "sees" (:505-579): the noisy latent, its one-step denoising, the predicted clean
        image, and the result of running the 50-step sampler (eta = 1) to the end from the nearest of its timesteps."""
        self.scheduler.set_timesteps(50)
        timesteps_dev = self.scheduler.timesteps.to(self.device)
        bs = min(self.cfg.max_items_eval, latents_noisy.shape[0]) if self.cfg.max_items_eval > 0 else latents_noisy.shape[0]
        large_enough_idxs = timesteps_dev.expand([bs, -1]) > t_orig[:bs].unsqueeze(-1)      # [bs, 50] > [bs, 1]
        idxs = torch.min(large_enough_idxs, dim=1)[1]
        t = timesteps_dev[idxs]
        fracs = list((t / self.scheduler.config.num_train_timesteps).cpu().numpy())
        imgs_noisy = self.decode_latents(latents_noisy[:bs]).permute(0, 2, 3, 1)
        latents_1step, pred_1orig = [], []
        for b in range(bs):
            step_output = self.scheduler.step(noise_pred[b:b + 1], t[b], latents_noisy[b:b + 1], eta=1, generator=generator)
            latents_1step.append(step_output["prev_sample"])
            pred_1orig.append(step_output["pred_original_sample"])
        latents_1step = torch.cat(latents_1step)
        pred_1orig = torch.cat(pred_1orig)
        imgs_1step = self.decode_latents(latents_1step).permute(0, 2, 3, 1)
        imgs_1orig = self.decode_latents(pred_1orig).permute(0, 2, 3, 1)
        latents_final = []
        for b, i in enumerate(idxs):
            latents = latents_1step[b:b + 1]
            sel = [b, b + len(idxs), b + 2 * len(idxs), b + 3 * len(idxs)] if use_perp_neg else [b, b + len(idxs)]
            text_emb = text_embeddings[sel, ...]
            neg_guid = neg_guidance_weights[b:b + 1] if use_perp_neg else None
            for tt in self.scheduler.timesteps[int(i) + 1:]:
                npred = self.get_noise_pred(latents, tt, text_emb, use_perp_neg, neg_guid)
                latents = self.scheduler.step(npred, tt, latents, eta=1, generator=generator)["prev_sample"]
            latents_final.append(latents)
        latents_final = torch.cat(latents_final)
        imgs_final = self.decode_latents(latents_final).permute(0, 2, 3, 1)
        return {"bs": bs, "noise_levels": fracs, "imgs_noisy": imgs_noisy, "imgs_1step": imgs_1step,
                "imgs_1orig": imgs_1orig, "imgs_final": imgs_final}

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        if self.cfg.grad_clip is not None:
            self.grad_clip_val = C(self.cfg.grad_clip, epoch, global_step)
        self.set_min_max_steps(min_step_percent=C(self.cfg.min_step_percent, epoch, global_step),
                               max_step_percent=C(self.cfg.max_step_percent, epoch, global_step))


def perpendicular_component(x, y):
    """The component of x perpendicular to y, per sample (threestudio/utils/ops.py:431-441)."""
    eps = torch.ones_like(x[:, 0, 0, 0]) * 1e-6
    return x - (torch.mul(x, y).sum(dim=[1, 2, 3]) / torch.maximum(torch.mul(y, y).sum(dim=[1, 2, 3]), eps)).view(-1, 1, 1, 1) * y


class PromptEmbeddings:
    """Stand-in for ``PromptProcessorOutput`` (prompt_processors/base.py:36-78): holds the four
    view-dependent (side/front/back/overhead) cond + uncond embeddings and returns
    ``[cond; uncond]`` ``[2B,77,1024]``.  The CLIP text encoder that fills it runs once at start-up
    in the reference and is out of scope; benchmarks use N(0,1) embeddings."""

    use_perp_neg = False
    # a * exp(-b r) + c  (prompt_processors/base.py:197-206)
    perp_neg_f_sb = (1, 0.5, -0.606)
    perp_neg_f_fsb = (1, 0.5, +0.967)
    perp_neg_f_fs = (4, 0.5, -2.426)
    perp_neg_f_sf = (4, 0.5, -2.426)

    def __init__(self, text_embeddings_vd, uncond_text_embeddings_vd, front_threshold=45.0, back_threshold=45.0,
                 overhead_threshold=60.0):
        self.text_embeddings_vd = text_embeddings_vd          # [4,77,1024]
        self.uncond_text_embeddings_vd = uncond_text_embeddings_vd
        self.text_embeddings = text_embeddings_vd[:1]
        self.uncond_text_embeddings = uncond_text_embeddings_vd[:1]
        self.front_threshold, self.back_threshold, self.overhead_threshold = front_threshold, back_threshold, \
            overhead_threshold

    @classmethod
    def random(cls, device, dtype=torch.float32, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        e = torch.randn(2, 4, 77, 1024, generator=g).to(device=device, dtype=dtype)
        return cls(e[0], e[1])

    def get_text_embeddings(self, elevation, azimuth, camera_distances, view_dependent_prompting=True):
        batch_size = elevation.shape[0]
        if view_dependent_prompting:
            # side=0 (default), front=1, back=2, overhead=3; later rules override earlier ones
            azi = (azimuth + 180) % 360 - 180  # shift_azimuth_deg
            idx = torch.zeros_like(elevation, dtype=torch.long)
            idx[(azi > -self.front_threshold) & (azi < self.front_threshold)] = 1
            idx[(azi > 180 - self.back_threshold) | (azi < -180 + self.back_threshold)] = 2
            idx[elevation > self.overhead_threshold] = 3
            dev = self.text_embeddings_vd.device
            if idx.device != dev:
                # cameras usually live on the host: a pageable host-to-device copy waits for everything queued on the stream (the
                # rasterizer forward of this very step) -- the one synchronising call torch's sync debug mode still found in the
                # SDS step in round 5; a pinned staging buffer + non_blocking copy does not wait
                if dev.type == "cuda" and not idx.is_cuda and sd21._PINNED_TABLES:
                    idx = idx.pin_memory()
                idx = idx.to(dev, non_blocking=True)
            text, uncond = self.text_embeddings_vd[idx], self.uncond_text_embeddings_vd[idx]
        else:
            text = self.text_embeddings.expand(batch_size, -1, -1)
            uncond = self.uncond_text_embeddings.expand(batch_size, -1, -1)
        return torch.cat([text, uncond], dim=0)

    def _direction_idx(self, elevation, azimuth):
        azi = (azimuth + 180) % 360 - 180
        idx = torch.zeros_like(elevation, dtype=torch.long)
        idx[(azi > -self.front_threshold) & (azi < self.front_threshold)] = 1
        idx[(azi > 180 - self.back_threshold) | (azi < -180 + self.back_threshold)] = 2
        idx[elevation > self.overhead_threshold] = 3
        return idx, azi

    def get_text_embeddings_perp_neg(self, elevation, azimuth, camera_distances, view_dependent_prompting=True):
        """``PromptProcessorOutput.get_text_embeddings_perp_neg`` (prompt_processors/base.py:80-160): per view the
        positive prompt interpolated between the front / side / back embeddings by azimuth, the unconditional one, and
        two negative prompts with weights  -(a exp(-b r) + c);  returns ([4B,77,1024] = pos | uncond | neg, [B,2])."""
        assert view_dependent_prompting, "Perp-Neg only works with view-dependent prompting"
        B = elevation.shape[0]
        idx, azi = self._direction_idx(elevation, azimuth)
        side, front, back, overhead = (self.text_embeddings_vd[i] for i in range(4))
        decay = lambda f, r: f[0] * torch.exp(-f[1] * r) + f[2]     # noqa: E731  shifted_expotional_decay
        pos, neg, wts, unc = [], [], [], []
        for i in range(B):
            d, a = int(idx[i]), azi[i]
            u = self.uncond_text_embeddings_vd[d]
            unc.append(u)
            if d == 3:                                   # overhead view: no negative direction
                pos.append(overhead)
                neg += [u, u]
                wts += [torch.zeros((), device=elevation.device), torch.zeros((), device=elevation.device)]
            elif torch.abs(a) < 90:                      # front-side interpolation (0 = side, 1 = front)
                r = 1 - torch.abs(a) / 90
                pos.append(r * front + (1 - r) * side)
                neg += [front, side]
                wts += [-decay(self.perp_neg_f_fs, r), -decay(self.perp_neg_f_sf, 1 - r)]
            else:                                        # side-back interpolation (0 = back, 1 = side)
                r = 2.0 - torch.abs(a) / 90
                pos.append(r * side + (1 - r) * back)
                neg += [side, front]
                wts += [-decay(self.perp_neg_f_sb, r), -decay(self.perp_neg_f_fsb, r)]
        emb = torch.cat([torch.stack(pos), torch.stack(unc), torch.stack(neg)], dim=0)
        w = torch.stack([torch.as_tensor(x, dtype=torch.float32) for x in wts]).to(elevation.device).reshape(B, 2)
        return emb, w
