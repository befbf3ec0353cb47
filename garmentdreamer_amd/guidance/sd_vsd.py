"""Variational-Score-Distillation guidance of the NeTF texture stage (BASELINE.json configs[4]).

Mirrors ``StableDiffusion.train_step`` / ``SpecifyGradient`` / ``encode_imgs``
(Garment_Deformer_NeTF/netf/guidance/sd_vsd_utils.py:15-28,131-218,274-282) and the LoRA training
step the trainer runs after each guidance step (Garment_Deformer_NeTF/netf/trainer.py:228-256):

    latents = VAE_enc(2x-1).sample() * 0.18215                    (with grad)
    t ~ U[20, 500]                    (t_range [0.02, 0.5], :39,162)
    eps_cfg = eps_uncond + s (eps_cond - eps_uncond)              (frozen UNet on [x_t; x_t], s = 7.5; :182-190)
    v_q = q_unet(x_t, t, text, c=pose, shading)                   (LoRA UNet, v-prediction)
    eps_q = sqrt(abar) v_q + sqrt(1-abar) x_t                     (:199-207)
    grad = (1 - abar) (eps_cfg - eps_q); loss = SpecifyGradient(latents, grad)      (:210-214)

As in the reference the batch size is 1 and the image is 512^2 (:144,146); embeddings are ordered
[cond; uncond] and CFG uses the usual ``uncond + s (cond - uncond)`` form (unlike the threestudio
guidance).  The UNets / VAE are the restatements in ``sd21.py`` (diffusers + hub weights are absent:
parity unpinned, see that module); bf16 on MI355X where the reference runs fp32 (``fp16=False``, :35).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import sd21
from .. import _runtime_env


import os as _os
_DRAIN = _os.environ.get("GD_VSD_DRAIN", "1") != "0"    # A/B toggle (round 5): drain the stream before the two training graphs


def _drain(device):
    """Wait for the stream before a training graph of the LoRA UNet is launched.  Measured, not derived: launched onto a busy stream
    the two graphs of the training pass (about 1500 nodes each) leave the GPU idle for about 1 ms per iteration more than launched
    onto a drained one (36.0-36.3 -> 35.1-35.4 ms per iteration, two interleaved runs on each of two boxes; the frozen networks'
    and the VAE's graphs show the opposite or nothing, and no runtime queue / kernarg-pool / signal-pool setting moves it:
    profiles/r05_vsd_drain_ab.txt).  The host has nothing else to do at these two points."""
    if _DRAIN and device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(device).synchronize()


class _DrainBeforeBackward(torch.autograd.Function):
    """Identity whose backward drains the stream first: it sits on the training UNet's output, so it runs right before the
    graphed backward pass of ``lu.backward()`` without the caller doing anything."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        _drain(g.device)
        return g


class SpecifyGradient(torch.autograd.Function):
    """sd_vsd_utils.py:15-28: forward returns sum(grad) as a dummy loss value, backward hands
    ``gt_grad / batch_size`` to the latents."""

    @staticmethod
    def forward(ctx, input_tensor, gt_grad):
        ctx.save_for_backward(gt_grad)
        return gt_grad.detach().sum().to(input_tensor.dtype)

    @staticmethod
    def backward(ctx, grad):
        (gt_grad,) = ctx.saved_tensors
        return gt_grad / len(gt_grad), None


class StableDiffusionVSD(nn.Module):
    def __init__(self, device, fp16: bool = True, t_range=(0.02, 0.5), unet: Optional[nn.Module] = None,
                 vae: Optional[nn.Module] = None, init_seed: int = 0, use_hip_graphs: bool = False,
                 fp8_unet: bool = False, fp8_calibration_steps: int = 3):
        super().__init__()
        # e4m3 MFMA convolutions in the iteration's three no-grad UNet forwards (BASELINE configs[4]; the reference is
        # fp32, sd_vsd_utils.py:35): the frozen UNet's two (cond / uncond, one batch) and the LoRA UNet's no-grad
        # forward; the LoRA UNet's TRAINING forward/backward stays bf16.  The first ``fp8_calibration_steps`` forwards
        # of each network run eagerly in bf16 and record the activation ranges (nn_ops.Fp8State).
        self.fp8_unet = bool(fp8_unet) and fp16 and torch.device(device).type == "cuda"
        self.fp8_calibration_steps = max(1, int(fp8_calibration_steps))
        self._fp8_calib = {}     # id(network) -> calibration forwards done
        # MI355X-side option (the reference has none): at batch 1 the iteration is ~5500 kernels of ~10 us and the
        # host cannot issue them fast enough; with it the frozen UNet forward, the LoRA UNet's no-grad forward, the
        # VAE encoder forward/backward and the LoRA UNet's training forward/backward replay as hipGraphs.
        self.use_hip_graphs = bool(use_hip_graphs) and torch.device(device).type == "cuda" and \
            _runtime_env.graph_replay_safe()
        self._graphs = {}
        self.device = torch.device(device)
        self.dtype = torch.bfloat16 if fp16 else torch.float32
        if unet is None:
            with torch.device(self.device):
                unet = sd21.init_random_(sd21.UNet2DConditionModel(), init_seed)
        if vae is None:
            with torch.device(self.device):
                vae = sd21.init_random_(sd21.AutoencoderKLEncoder(), init_seed + 1)
        self.unet = unet.to(device=self.device, dtype=self.dtype).to(memory_format=torch.channels_last).eval()
        self.vae = vae.to(device=self.device, dtype=self.dtype).to(memory_format=torch.channels_last).eval()
        for p in list(self.unet.parameters()) + list(self.vae.parameters()):
            p.requires_grad_(False)
        if self.fp8_unet:
            self.unet.enable_fp8()
        self.scheduler = sd21.DDIMScheduler()
        self.num_train_timesteps = self.scheduler.config.num_train_timesteps
        self.min_step = int(self.num_train_timesteps * t_range[0])
        self.max_step = int(self.num_train_timesteps * t_range[1])
        self.alphas = self.scheduler.alphas_cumprod.to(self.device)
        self.embeddings = {}

    def set_text_embeds(self, pos, neg, front=None, side=None, back=None):
        """The reference fills these with the CLIP text encoder (get_text_embeds, :81-91); out of
        scope here, so callers provide [1,77,1024] tensors (random for benchmarks)."""
        self.embeddings = {"pos": pos, "neg": neg, "front": front if front is not None else pos,
                           "side": side if side is not None else pos, "back": back if back is not None else pos}

    # ---- hipGraph replay (same scheme as StableDiffusionGuidance._graphed_unet / _graphed_vae_moments) ----
    def _graphs_failed(self, err):
        import warnings
        warnings.warn(f"hipGraph capture failed ({err}); continuing with eager kernel launches")
        self.use_hip_graphs = False
        self._graphs.clear()
        torch.cuda.synchronize()
        from .. import nn_ops
        nn_ops.reset_workspaces()

    def _replay_nograd(self, key, fn, *tensors):
        """``fn(*tensors)`` under no_grad, captured once per (key, shapes) and replayed on static copies."""
        key = (key,) + tuple(tuple(t.shape) for t in tensors)
        entry = self._graphs.get(key)
        if entry is None:
            static = [t.clone() for t in tensors]
            dev = static[0].device
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):
                    fn(*static)
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                out = fn(*static)
            entry = self._graphs[key] = (g, static, out)
        g, static, out = entry
        for st, t in zip(static, tensors):
            st.copy_(t)
        g.replay()
        return out.clone()

    def _graphed_module(self, key, make_module, *tensors):
        """Autograd-aware forward/backward graph pair of ``make_module()`` (torch.cuda.make_graphed_callables)."""
        key = (key,) + tuple(tuple(t.shape) for t in tensors)
        fn = self._graphs.get(key)
        if fn is None:
            sample = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in tensors)
            module = make_module()
            # gradient sinks (flat_adam.FlatAdam: the LoRA backward kernels ADD the adapter gradients into them) would
            # collect the gradients of the warm-up passes torch runs before the capture: put their contents back afterwards
            bases = {}
            for prm in module.parameters():
                sink = getattr(prm, "_gd_grad_sink", None)
                if sink is not None:
                    base = sink._base if sink._base is not None else sink
                    bases[id(base)] = base
            saved = [(b, b.clone()) for b in bases.values()]
            fn = self._graphs[key] = torch.cuda.make_graphed_callables(module, sample, allow_unused_input=True)
            with torch.no_grad():
                for b, c in saved:
                    b.copy_(c)
        return fn(*tensors)

    def encode_imgs(self, imgs, vae_noise=None):
        imgs = 2 * imgs - 1
        x = imgs.to(self.dtype)
        posterior = None
        if self.use_hip_graphs and x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
            vae = self.vae

            class _Moments(nn.Module):
                def forward(self, z):
                    return vae.quant_conv(vae.encoder(z))

            try:
                posterior = sd21.DiagonalGaussianDistribution(
                    self._graphed_module("vae", _Moments, x.contiguous(memory_format=torch.channels_last)))
            except RuntimeError as e:
                self._graphs_failed(e)
        if posterior is None:
            posterior = self.vae.encode(x).latent_dist
        return posterior.sample(vae_noise) * self.vae.config.scaling_factor

    def _fp8_calibrating(self, net) -> bool:
        """True while ``net`` (a UNet with an Fp8State) still has to run eager bf16 calibration forwards; counts one."""
        st = getattr(net, "fp8", None)
        if st is None or st.mode == "run":
            return False
        n = self._fp8_calib.get(id(net), 0) + 1
        self._fp8_calib[id(net)] = n
        if n > self.fp8_calibration_steps:
            st.mode = "run"
            return False
        return True

    def _frozen_unet(self, x, t, ctx):
        x, ctx = x.to(self.dtype), ctx.to(self.dtype)
        if self._fp8_calibrating(self.unet):
            return self.unet(x, t, encoder_hidden_states=ctx)
        if self.use_hip_graphs and x.is_cuda:
            try:
                return self._replay_nograd("unet", lambda a, b, c: self.unet(a, b, encoder_hidden_states=c), x, t, ctx)
            except RuntimeError as e:
                self._graphs_failed(e)
        return self.unet(x, t, encoder_hidden_states=ctx)

    def _q_nograd(self, q_unet, x, t, text, pose, shading):
        inner = getattr(q_unet, "unet", None)
        if self.fp8_unet and inner is not None and hasattr(inner, "enable_fp8"):
            if inner.fp8 is None and x.is_cuda and next(inner.parameters()).dtype == torch.bfloat16:
                inner.enable_fp8()
            if self._fp8_calibrating(inner):
                return q_unet(x, t, text, c=pose, shading=shading)
        if self.use_hip_graphs and x.is_cuda:
            try:
                return self._replay_nograd(("q", id(q_unet), shading),
                                           lambda a, b, c, d: q_unet(a, b, c, c=d, shading=shading), x, t, text, pose)
            except RuntimeError as e:
                self._graphs_failed(e)
        return q_unet(x, t, text, c=pose, shading=shading)

    def _q_train(self, q_unet, x, t, text, pose, shading):
        if self.use_hip_graphs and x.is_cuda and isinstance(q_unet, nn.Module):
            class _Q(nn.Module):
                def __init__(self):
                    super().__init__()
                    self.q = q_unet

                def forward(self, a, b, c, d):
                    return self.q(a, b, c, c=d, shading=shading)

            try:
                return self._graphed_module(("q_train", id(q_unet), shading), _Q, x, t, text, pose)
            except RuntimeError as e:
                self._graphs_failed(e)
        return q_unet(x, t, text, c=pose, shading=shading)

    def train_step(self, pred_rgb, guidance_scale=7.5, q_unet=None, pose=None, shading=None, as_latent=False,
                   t5=False, hors=None, noise=None, timesteps=None, vae_noise=None):
        batch_size = pred_rgb.shape[0]
        assert batch_size == 1
        assert pred_rgb.shape[2] == pred_rgb.shape[3] == 512
        assert not as_latent
        latents = self.encode_imgs(pred_rgb, vae_noise).float()
        if timesteps is not None:
            t = timesteps.to(self.device).long()
        elif t5:
            t = torch.randint(self.min_step, 500 + 1, [1], dtype=torch.long, device=self.device)
        else:
            t = torch.randint(self.min_step, self.max_step + 1, (batch_size,), dtype=torch.long, device=self.device)
        with torch.no_grad():
            if noise is None:
                noise = torch.randn_like(latents)
            latents_noisy = self.scheduler.add_noise(latents, noise, t)
            latent_model_input = torch.cat([latents_noisy] * 2)
            tt = torch.cat([t] * 2)
            if hors is None:
                embeddings = torch.cat([self.embeddings["pos"].expand(batch_size, -1, -1),
                                        self.embeddings["neg"].expand(batch_size, -1, -1)])
            else:
                def _dir(h):
                    return "front" if abs(h) < 60 else ("side" if abs(h) < 120 else "back")
                embeddings = torch.cat([self.embeddings[_dir(h)] for h in hors] +
                                       [self.embeddings["neg"].expand(batch_size, -1, -1)])
            noise_pred = self._frozen_unet(latent_model_input, tt, embeddings).float()
            noise_pred_cond, noise_pred_uncond = noise_pred.chunk(2)
            noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_cond - noise_pred_uncond)
            if q_unet is None or pose is None:
                raise NotImplementedError("VSD needs the LoRA UNet and a pose (sd_vsd_utils.py:192-197)")
            v_q = self._q_nograd(q_unet, latents_noisy, t, self.embeddings["pos"].expand(batch_size, -1, -1).contiguous(),
                                 pose, shading or "albedo").float()
            a = self.alphas[t].view(-1, 1, 1, 1)
            noise_pred_q = a.sqrt() * v_q + (1 - a).sqrt() * latents_noisy   # v -> eps
        w = (1 - self.alphas[t]).view(batch_size, 1, 1, 1)
        grad = torch.nan_to_num(w * (noise_pred - noise_pred_q))
        loss = SpecifyGradient.apply(latents, grad)
        pseudo_loss = torch.mul((w * noise_pred).detach(), latents.detach()).detach().sum()
        return loss, pseudo_loss, latents

    def lora_train_loss(self, q_unet, latents, pose, shading="albedo", unet_bs=1, v_pred=True, uncond_p=0.1,
                        timesteps=None, noise=None, drop_pose: Optional[bool] = None):
        """One denoising-loss evaluation for the LoRA UNet (trainer.py:228-256): MSE to the velocity
        (or noise) target on the current latents; the caller backprops and steps its optimizer."""
        with torch.no_grad():
            latents_clean = latents.detach().expand(unet_bs, *latents.shape[1:]).contiguous()
            pose_b = pose.expand(unet_bs, 16).contiguous()
            if drop_pose is None:
                drop_pose = bool(torch.rand(()) < uncond_p)
            if drop_pose:
                pose_b = torch.zeros_like(pose_b)
            if timesteps is None:
                timesteps = torch.randint(0, 1000, (unet_bs,), device=self.device).long()
            if noise is None:
                noise = torch.randn(latents_clean.shape, device=self.device)
            latents_noisy = self.scheduler.add_noise(latents_clean, noise, timesteps)
            target = self.scheduler.get_velocity(latents_clean, noise, timesteps) if v_pred else noise
        graphed = self.use_hip_graphs and latents_noisy.is_cuda
        if graphed:
            _drain(latents_noisy.device)
        out = self._q_train(q_unet, latents_noisy, timesteps,
                            self.embeddings["pos"].expand(unet_bs, -1, -1).contiguous(), pose_b, shading)
        if graphed and out.requires_grad:
            out = _DrainBeforeBackward.apply(out)
        return F.mse_loss(out.float(), target)


class LoraUnet(nn.Module):
    """trainer.py:107-116: wraps the adapter UNet and repeats the single text embedding over the batch."""

    def __init__(self, unet: sd21.LoraUNet2DConditionModel):
        super().__init__()
        self.unet = unet
        self.sample_size = 64
        self.in_channels = 4

    def forward(self, x, t, text_embeddings, c=None, shading="albedo"):
        textemb = text_embeddings.expand(x.shape[0], -1, -1) if text_embeddings.shape[0] == 1 else text_embeddings
        return self.unet(x, t, encoder_hidden_states=textemb, c=c, shading=shading)
