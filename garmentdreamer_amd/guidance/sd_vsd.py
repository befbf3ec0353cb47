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
_CONCURRENT = _os.environ.get("GD_VSD_CONCURRENT", "1") != "0"   # A/B toggle (round 5): frozen UNet and LoRA no-grad forward on two streams
_THREE_STREAMS = _os.environ.get("GD_VSD_THREE_STREAMS", "1") != "0"   # A/B toggle (round 5): the training pass on a stream of its own (=0: it shares the no-grad forward's)
_TRAIN_STREAM = _os.environ.get("GD_VSD_TRAIN_STREAM", "1") != "0"   # A/B toggle (round 5): the LoRA training pass on the side stream
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
    def lora_stream(self):
        """``with guidance.lora_stream(): lu.backward(); lora_optimizer.step()`` -- the LoRA UNet's backward pass and optimizer step on
        the stream its forward passes already run on, WITHOUT the caller's stream waiting for them: the next iteration's VAE
        encoder and frozen UNet (which do not depend on the adapters) then run beside them, and ``train_step`` joins exactly where
        it needs the updated adapters (its LoRA forward runs on this stream, behind the optimizer step).  Entering waits for what
        the caller queued so far (``zero_grad``); leaving does NOT join: code that touches the adapters or their gradients outside
        ``train_step`` / ``lora_train_loss`` calls ``join_lora_stream()`` first.  A no-op context on the CPU / without hipGraphs."""
        import contextlib
        if not (_CONCURRENT and _TRAIN_STREAM and self.use_hip_graphs and torch.cuda.is_available()
                and torch.device(self.device).type == "cuda"):
            return contextlib.nullcontext()
        dev = torch.device(self.device)
        side = self._side_stream(dev, train=True)
        side.wait_stream(torch.cuda.current_stream(dev))
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:          # the leaves were created on the caller's stream and accumulate on this one, on purpose
            quiet(False)
        guidance = self

        class _Ctx:
            def __enter__(self_c):
                self_c.inner = torch.cuda.stream(side)
                return self_c.inner.__enter__()

            def __exit__(self_c, *exc):
                # the next no-grad LoRA forward (on the other LoRA stream) starts behind the optimizer step
                guidance._ev_weights = torch.cuda.Event()
                guidance._ev_weights.record(side)
                return self_c.inner.__exit__(*exc)
        return _Ctx()

    def join_lora_stream(self):
        """The caller's stream waits for everything queued on the LoRA stream(s) (see ``lora_stream``)."""
        for name in ("_side", "_side_train"):
            st = getattr(self, name, None)
            if st is not None:
                torch.cuda.current_stream(st.device).wait_stream(st)

    def _side_stream(self, device, train: bool = False):
        """The LoRA no-grad stream, or (``train``) the stream of the training pass (forward, backward, optimizer step): two streams, so
        that the training forward -- which needs the latents only -- does not queue behind the no-grad forward of the same iteration
        (29.5 -> 24.9 ms per iteration; GD_VSD_THREE_STREAMS=0: one LoRA stream for both)."""
        name = "_side_train" if train and _THREE_STREAMS else "_side"
        st = getattr(self, name, None)
        if st is None or st.device != device:
            # default priority: a high-priority training stream (priority=-1) measured 61 ms per iteration against 25 (round 5).
            # Created where first needed, nothing cleverer: HIP deals streams to its few hardware queues in creation order, and
            # two attempts of round 6 to pick these streams "better" (re-drawing until distinct from the capture streams; both
            # drawn back to back) cost 24.9 -> 28.3-29.5 ms per iteration -- the streams then shared a hardware queue with
            # something they should overlap (GPU_MAX_HW_QUEUES=8: 52.9 ms).  A side stream never captures, so it may coincide
            # with a capture stream without two graphs sharing a library workspace.
            st = torch.cuda.Stream(device=device)
            setattr(self, name, st)
        return st

    def _capture_stream(self, device):
        """A capture stream no other graph of this guidance uses: the library GEMMs'
        workspace is keyed by the stream a call was CAPTURED on, and graphs that replay concurrently must not share one.
        torch.cuda.Stream() hands out a pool of 32 streams per device round-robin, so after enough graph keys two 'new' streams
        are the same stream -- draw until the handle is unused (and say so if the pool is exhausted)."""
        used = getattr(self, "_capture_handles", None)
        if used is None:
            used = self._capture_handles = set()
        for _ in range(64):
            st = torch.cuda.Stream(device=device)
            if st.cuda_stream not in used and st.cuda_stream != torch.cuda.current_stream(device).cuda_stream:
                used.add(st.cuda_stream)
                return st
        raise RuntimeError("StableDiffusionVSD: no unused capture stream left (torch's pool of 32 streams per device is shared by "
                           "every graph key of this guidance): concurrently replayed graphs would share a library-GEMM workspace")

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
            from .. import nn_ops
            with nn_ops.workspace_tag(key):        # its own GroupNorm accumulators: graphs of two networks may replay concurrently
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(2):
                        fn(*static)
                torch.cuda.current_stream(dev).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # ... and its own CAPTURE stream: the library GEMMs' workspace (stream-K partial tiles) is keyed by the stream the
                # call was captured on, so two graphs captured on torch's default capture stream would share one
                cap = self._capture_stream(dev)
                with torch.cuda.graph(g, stream=cap), torch.no_grad():
                    out = fn(*static)
            entry = self._graphs[key] = (g, static, out, cap)
        g, static, out = entry[:3]
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
            # own capture stream (= own library-GEMM workspace) and own GroupNorm accumulators, as in _replay_nograd: the VAE's
            # backward graph and the LoRA UNet's training graphs may replay concurrently.  make_graphed_callables captures on
            # torch.cuda.graph's class-level default capture stream: swapped for the duration of the call
            from .. import nn_ops
            cap = self._capture_stream(tensors[0].device)
            prev_cap = torch.cuda.graph.default_capture_stream
            torch.cuda.graph.default_capture_stream = cap
            try:
                with nn_ops.workspace_tag(key):
                    fn = self._graphs[key] = torch.cuda.make_graphed_callables(module, sample, allow_unused_input=True)
            finally:
                torch.cuda.graph.default_capture_stream = prev_cap
            self._capture_streams = getattr(self, "_capture_streams", []) + [cap]
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
        if latents.is_cuda and not torch.cuda.is_current_stream_capturing():
            # lora_train_loss needs the latents and nothing that comes after them (not the frozen UNet, not the VAE backward the
            # caller runs in between): it waits for THIS event on its own stream
            self._ev_latents = torch.cuda.Event()
            self._ev_latents.record(torch.cuda.current_stream(latents.device))
            # the event orders the training stream behind THESE latents only: lora_train_loss checks it is handed them
            self._ev_latents_id = (latents.data_ptr(), latents._version)
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
            if q_unet is None or pose is None:
                raise NotImplementedError("VSD needs the LoRA UNet and a pose (sd_vsd_utils.py:192-197)")
            text_q = self.embeddings["pos"].expand(batch_size, -1, -1).contiguous()
            overlap = _CONCURRENT and self.use_hip_graphs and latents_noisy.is_cuda and \
                not torch.cuda.is_current_stream_capturing()
            if overlap:
                # the frozen UNet (2 latents) and the LoRA UNet's no-grad forward (1 latent) are independent and each far too
                # small to fill the chip (launch-bound, ~700 kernels of ~7 us): the second graph replays on the LoRA side stream
                # beside the first (round 5; GD_VSD_CONCURRENT=0 = one after the other).  The frozen UNet is queued FIRST: the side
                # stream may still hold the previous iteration's LoRA backward pass and optimizer step (lora_stream()), and a
                # graph launch keeps the host for about as long as the work in front of it on its stream takes.
                cur = torch.cuda.current_stream(latents_noisy.device)
                side = self._side_stream(latents_noisy.device)
                inputs_ready = torch.cuda.Event()
                inputs_ready.record(cur)
            noise_pred = self._frozen_unet(latent_model_input, tt, embeddings).float()
            noise_pred_cond, noise_pred_uncond = noise_pred.chunk(2)
            noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_cond - noise_pred_uncond)
            if overlap:
                side.wait_event(inputs_ready)
                ev_w = getattr(self, "_ev_weights", None)
                if ev_w is not None:           # lora_stream(): the optimizer step of the previous iteration, wherever it ran
                    side.wait_event(ev_w)
                with torch.cuda.stream(side):
                    v_q = self._q_nograd(q_unet, latents_noisy, t, text_q, pose, shading or "albedo").float()
                cur.wait_stream(side)
                v_q.record_stream(cur)
            else:
                v_q = self._q_nograd(q_unet, latents_noisy, t, text_q, pose, shading or "albedo").float()
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
        import contextlib
        ctx = contextlib.nullcontext()
        if _CONCURRENT and _TRAIN_STREAM and self.use_hip_graphs and latents.is_cuda and not torch.cuda.is_current_stream_capturing():
            # The training pass depends on the latents only -- not on the frozen UNet's score, not on the VAE backward pass the
            # caller has queued since train_step.  It runs on the LoRA side stream (round 5): forward beside the VAE backward;
            # autograd runs its backward on the stream of its forward and joins the caller's stream when backward() returns.
            cur = torch.cuda.current_stream(latents.device)
            side = self._side_stream(latents.device, train=True)
            ev = getattr(self, "_ev_latents", None)
            self._ev_latents = None            # good for ONE call: a second training pass after the same train_step (trainer.py's
            #                                    K loop) must also see the optimizer step the caller ran in between on ITS stream
            same = getattr(self, "_ev_latents_id", None) == (latents.data_ptr(), latents._version)
            if ev is None or not same or timesteps is not None or noise is not None:
                # caller-made inputs (tests) / repeated call / latents that are NOT train_step's output (a replay buffer, re-encoded
                # latents produced later on the caller's stream): everything the caller queued so far
                side.wait_stream(cur)
            else:
                side.wait_event(ev)
            ctx = torch.cuda.stream(side)
        with ctx:
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
            loss = F.mse_loss(out.float(), target)
        if not isinstance(ctx, contextlib.nullcontext):
            cur.wait_stream(side)            # the caller reads / differentiates the loss on ITS stream (the VAE backward queued there
            loss.record_stream(cur)          # before this call still ran beside the forward pass)
        return loss


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
