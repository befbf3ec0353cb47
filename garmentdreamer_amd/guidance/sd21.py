"""Stable-Diffusion-2.1 UNet, VAE encoder and DDIM noise schedule, restated for PyTorch-ROCm.

The reference obtains these from the un-vendored dependency ``diffusers==0.19.0``
(Garment_3DGS/requirements.txt:12; call sites Garment_3DGS/threestudio/models/guidance/
stable_diffusion_guidance.py:66-69,120-131,153-157,165-166) with weights fetched from the HF hub at
run time.  Neither the package nor the weights exist in this environment, so the public SD-2.1
architecture is restated here and run with random-init weights; PARITY IS UNPINNED for this part
(SURVEY 8a row a21, 8c): tests check self-consistency (bf16 path vs fp32 path, analytic scheduler
constants, shape/param-count facts of the published config) and nothing else can be checked.

Architecture facts (public ``stabilityai/stable-diffusion-2-1-base`` config.json; skeleton visible
in-tree at Garment_Deformer_NeTF/netf/vsd/lora_unet.py:119-160):
  UNet: in/out 4, block_out_channels (320,640,1280,1280), layers_per_block 2, down = 3x
        CrossAttnDownBlock2D + DownBlock2D, mid = UNetMidBlock2DCrossAttn, up = UpBlock2D + 3x
        CrossAttnUpBlock2D, attention heads (5,10,20,20) of dim 64, cross_attention_dim 1024,
        use_linear_projection, GroupNorm(32), SiLU, time embedding 320 -> 1280.
  VAE encoder: block_out_channels (128,256,512,512), layers_per_block 2, one single-head mid
        attention (d=512), latent_channels 4 (conv_out 8 = mean|logvar), scaling_factor 0.18215.
  Scheduler: scaled_linear betas in [0.00085, 0.012], 1000 steps.
Module/parameter names follow diffusers' state-dict keys so real checkpoints load with
``load_state_dict`` (``load_diffusers_weights``).

Layout choices for MI355X: activations are kept channels_last (NHWC) so the 3x3 convolutions run
as implicit GEMMs over contiguous K = 9*C_in; attention goes through
``F.scaled_dot_product_attention`` (flash-style, no N^2 tensor in HBM); weights and activations
are bf16 (the reference uses fp16), normalisation statistics and softmax accumulate in fp32.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nn_ops import (add_layer_norm, attention_d64, attention_d64_supported, attention_d64_vt, conv1x1, conv3x3, conv3x3_s2, conv3x3_s2_supported, conv3x3_small_cin,
                      conv3x3_supported, geglu, gn_conv3x3, gn_conv3x3_supported, gn_conv_prefers_fused, group_norm_silu,
                      resnet_block_frozen, resnet_block_frozen_supported, upsample2x_conv3x3,
                      upsample2x_conv3x3_supported, linear_320, route_rows, route_scale,
                      linear_320_geglu, linear_320_supported, gemm_geglu, gemm_supported)


import os as _os

_FUSED_QKV = _os.environ.get("GD_FUSED_QKV", "1") != "0"   # A/B toggle of the fused self-attention projection
_VT_GEMM = _os.environ.get("GD_VT_GEMM", "1") != "0"         # A/B toggle: V^T from a GEMM instead of the transposing pre-pass
_LORA_FUSED = _os.environ.get("GD_LORA_FUSED", "1") != "0"   # A/B toggle: own rank-4 LoRA kernels (fwd + bwd) instead of torch ops
_VAE_ATTN_NODE = _os.environ.get("GD_VAE_ATTN_NODE", "1") != "0"  # A/B toggle: VAE mid attention as one autograd node with own softmax
_CTX_VT = _os.environ.get("GD_CTX_VT", "1") != "0"           # A/B toggle: cross-attention V^T of all layers from one GEMM (round 5)
_LORA_LINEAR = _os.environ.get("GD_LORA_LINEAR", "1") != "0"  # A/B toggle: adapted projection as ONE autograd node (nn_ops.lora_linear)
_CONV_BIAS_GRAD = _os.environ.get("GD_CONV_BIAS_GRAD", "1") != "0"  # A/B toggle (round 5): trainable per-image bias inside the conv node
_PINNED_TABLES = _os.environ.get("GD_PAGEABLE_COPIES", "0") != "1"  # A/B toggle (round 5): =1 restores the per-call pageable host-to-device copies (each a full sync)
_TEMB_TRAIN_CAT = _os.environ.get("GD_TEMB_TRAIN_CAT", "1") != "0"  # A/B toggle (round 5): the 22 time_emb_proj of a TRAINING pass as one GEMM
from .. import nn_ops  # noqa: E402
_FP8_ACTIVE = [None]   # the nn_ops.Fp8State of the UNet whose no-grad forward is running (set by its forward)


def _gn(norm: nn.GroupNorm, x, silu: bool):
    """GroupNorm (+SiLU): fused NHWC HIP kernel on the GPU (nn_ops), torch ops on CPU."""
    return group_norm_silu(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu)


def _conv3(conv: nn.Conv2d, x, image_bias=None, residual=None):
    """3x3/s1/p1 convolution with the bias (or a per-image bias [N,Cout]) and an optional residual
    fused into the MFMA kernel's epilogue (nn_ops.conv3x3; small feature maps run split over the nine taps);
    falls back to torch's conv for layers the kernel does not cover (Cin % 64 != 0, fp32, CPU)."""
    bias = conv.bias if image_bias is None else image_bias
    frozen = not conv.weight.requires_grad and (conv.bias is None or not conv.bias.requires_grad)
    if frozen and conv3x3_supported(x, conv.weight) and conv.stride == (1, 1) and conv.padding == (1, 1):
        if bias is not None and bias.requires_grad and not _CONV_BIAS_GRAD:   # round-4 form (same-box A/B): bias and residual as aten adds
            y = conv3x3(x, conv.weight, None, None) + (bias[:, :, None, None] if bias.dim() == 2 else bias[None, :, None, None])
            return y if residual is None else y + residual
        # LoRA training: the time / camera embedding bias needs a gradient -- the node returns it (the sum of dy over the pixels)
        return conv3x3(x, conv.weight, bias, residual)
    nn_ops._note_fallback("sd21._conv3", x, "needs frozen bf16 channels_last weights, stride 1, pad 1, Cin % 64 == 0")
    y = F.conv2d(x, conv.weight, None if image_bias is not None else conv.bias, conv.stride, conv.padding)
    if image_bias is not None:
        y = y + image_bias[:, :, None, None]
    return y if residual is None else y + residual


import os as _os0
_LIN320 = _os0.environ.get("GD_LINEAR320", "1") != "0"      # =0: library GEMM (same-box A/B in tools/, never set in tests)
_OWN_GEMM = _os0.environ.get("GD_OWN_GEMM", "1") != "0"     # =0: GEGLU projections with K != 320 on the library GEMM + geglu_kernel
_OWN_GEMM_MIN_ROWS = 1024                                    # below: the 256-row tiles cannot fill the chip (library + geglu_kernel)


def _lib_linear(x, w, b=None):
    """``F.linear`` on the library GEMM (hipBLASLt).  The library picks its tile / split-K from M, and the summation
    order of a row can depend on the tile it falls in, so a rank holding 1/k of the views would not reproduce the
    single-rank run of the whole batch: under batch-invariant selection (nn_ops.set_route_scale(k),
    SDSLoop(batch_invariant=True)) the GEMM runs on the single-rank row set with this rank's rows at their global
    positions and zeros elsewhere (nn_ops.route_rows; tools/batch_invariance_probe.py, tools/guidance_invariance.py:
    the 16x16-token GEGLU projection is the product of the UNet forward whose bits change between 16 and 8 latents)."""
    if route_scale() > 1 and x.is_cuda:
        padded, take = route_rows(x.reshape(-1, x.shape[-1]))
        return take(F.linear(padded, w, b)).reshape(*x.shape[:-1], w.shape[0])
    return F.linear(x, w, b)


def _lib_addmm(c, a, w_t):
    """``torch.addmm(c, a, w_t)`` with the same row placement under batch-invariant selection (see _lib_linear)."""
    if route_scale() > 1 and a.is_cuda:
        pa, take = route_rows(a)
        pc, _ = route_rows(c)
        return take(torch.addmm(pc, pa, w_t))
    return torch.addmm(c, a, w_t)


def _lin(mod: nn.Linear, x, bias=None):
    """``mod(x)`` (with ``bias`` in place of ``mod.bias`` if given).  Frozen products with K = 320 on long row sets -- to_q
    of the cross-attention, to_out.0, proj_in, proj_out (N = 320) and the GEGLU projection (N = 2560) of the 64x64-token
    transformer blocks -- run on the weights-in-registers streaming kernel (nn_ops.linear_320: they are HBM streams;
    1.4x / 1.1x the library GEMM on MI355X, tools/linear320_bench.py)."""
    w = mod.weight
    b = mod.bias if bias is None else bias
    if _LIN320 and x.is_cuda and not torch.is_grad_enabled() and not w.requires_grad and w.shape[1] == 320 and \
            linear_320_supported(x, w, b):
        return linear_320(x, w, b)
    return _lib_linear(x, w, b)


def _gn_conv3(norm: nn.GroupNorm, conv: nn.Conv2d, x, image_bias=None, residual=None):
    """``conv(silu(norm(x)))`` of a ResnetBlock.  Large feature maps (the VAE encoder's 512^2 .. 128^2 levels) run
    as ONE patch-staged kernel that normalises in its activation loader (nn_ops.gn_conv3x3: 1.05-1.26x faster
    than GroupNorm kernel + convolution on MI355X, tools/gn_conv_bench.py); on 64^2 and smaller maps the
    GroupNorm kernel + plain convolution (LDS-DMA patch kernel or implicit GEMM, chosen by the library) is faster."""
    frozen = not conv.weight.requires_grad and (conv.bias is None or not conv.bias.requires_grad)
    frozen = frozen and (image_bias is None or not image_bias.requires_grad)
    st = _FP8_ACTIVE[0]
    if st is not None and frozen and not torch.is_grad_enabled() and st.wants(conv, x):
        # e4m3 path of the no-grad UNet forward (nn_ops.Fp8State): calibration runs this site in bf16 and records
        # the activation range, afterwards GroupNorm+SiLU writes e4m3 and the convolution runs on the fp8 kernel
        bias = conv.bias if image_bias is None else image_bias
        if st.mode == "run" and id(conv) in st.amax:
            return st.gn_conv(norm, conv, x, bias, residual)
        act = _gn(norm, x, True)
        if st.mode == "calibrate":
            st.observe(conv, act)
        return _conv3(conv, act, image_bias=image_bias, residual=residual)
    if x.is_cuda and frozen and gn_conv_prefers_fused(x, conv.out_channels) and \
            gn_conv3x3_supported(x, norm.weight, conv.weight):
        return gn_conv3x3(x, norm.weight, norm.bias, norm.num_groups, norm.eps, True, conv.weight,
                          conv.bias if image_bias is None else image_bias, residual)
    return _conv3(conv, _gn(norm, x, True), image_bias=image_bias, residual=residual)


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------


class TembProjections:
    """``time_emb_proj(silu(temb)) + conv1.bias`` of every ResnetBlock2D of a frozen UNet from ONE GEMM over the
    row-concatenated projection weights: the 22 blocks of SD-2.1 otherwise launch 22 x (silu, GEMM, add) on a
    [B, 1280] tensor -- 64 launches of ~5 us that the 1-view-per-GPU step cannot hide.  ``image_bias[id(block)]``
    is a [B, C_out] column slice (row stride = the total width); the conv kernels take the stride."""

    __slots__ = ("temb", "image_bias")

    def __init__(self, temb, image_bias):
        self.temb, self.image_bias = temb, image_bias


class ContextProjections:
    """``to_k(context)`` / ``to_v(context)`` of every cross-attention of a frozen UNet from ONE GEMM: the text
    embeddings do not depend on the sample, so the 16 layers' 32 projections of the [B, 77, 1024] context are one
    [B*77, 1024] x [1024, 24960] product whose column slices feed the attentions as strided views."""

    __slots__ = ("context", "kv", "vt")

    def __init__(self, context, kv, vt=None):
        # vt[id(attn)]: V^T of the zero-padded context, a [B, C, 128] channel slice of ONE W_v_cat . context^T product
        # (round 5): the attention kernel consumes V transposed, and the per-layer transposing pre-pass (15 launches per
        # UNet forward) is gone
        self.context, self.kv, self.vt = context, kv, (vt or {})

    @property
    def shape(self):
        return self.context.shape


class ResnetBlock2D(nn.Module):
    def __init__(self, in_ch: int, out_ch: int, temb_ch: Optional[int], eps: float = 1e-5, groups: int = 32):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_ch, eps=eps)
        self.conv1 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, out_ch) if temb_ch is not None else None
        self.norm2 = nn.GroupNorm(groups, out_ch, eps=eps)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_ch, out_ch, 1) if in_ch != out_ch else None

    def forward(self, x, temb=None, next_norm=None):
        if resnet_block_frozen_supported(x, self):   # VAE encoder inside the SDS graph: one autograd node per block
            return resnet_block_frozen(x, self, next_norm)
        image_bias = None
        if isinstance(temb, TembProjections):   # projected for every block at once (UNet2DConditionModel.forward)
            image_bias = temb.image_bias.get(id(self))
            temb = temb.temb
        if image_bias is None and self.time_emb_proj is not None:
            # conv bias + time-embedding projection = one per-image bias
            image_bias = self.time_emb_proj(F.silu(temb)) + self.conv1.bias
        h = _gn_conv3(self.norm1, self.conv1, x, image_bias=image_bias)
        if self.conv_shortcut is not None:
            x = conv1x1(x, self.conv_shortcut.weight, self.conv_shortcut.bias)
        return _gn_conv3(self.norm2, self.conv2, h, residual=x)


class LoRALinearLayer(nn.Module):
    """diffusers 0.19 ``LoRALinearLayer``: down (in -> rank, N(0, 1/rank)) then up (rank -> out, zeros)."""

    def __init__(self, in_features: int, out_features: int, rank: int = 4):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, x):
        return self.up(self.down(x.to(self.down.weight.dtype))).to(x.dtype)


class Attention(nn.Module):
    """Multi-head attention with diffusers' parameter names (to_q/to_k/to_v/to_out.0).  ``add_lora``
    attaches the four low-rank adapters of diffusers' ``LoRAAttnProcessor`` (used by the NeTF stage's
    trainable "q" UNet: Garment_Deformer_NeTF/netf/trainer.py:79-101)."""

    def __init__(self, query_dim: int, heads: int, dim_head: int, cross_dim: Optional[int] = None,
                 qkv_bias: bool = False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=qkv_bias)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=qkv_bias)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=qkv_bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim)])
        self.lora = None
        self.lora_scale = 1.0

    def add_lora(self, rank: int = 4):
        q_in, kv_in, inner = self.to_q.in_features, self.to_k.in_features, self.to_q.out_features
        self.lora = nn.ModuleDict({
            "to_q_lora": LoRALinearLayer(q_in, inner, rank), "to_k_lora": LoRALinearLayer(kv_in, inner, rank),
            "to_v_lora": LoRALinearLayer(kv_in, inner, rank),
            "to_out_lora": LoRALinearLayer(inner, self.to_out[0].out_features, rank)})
        return self.lora

    def forward(self, x, context=None):
        B, N, _ = x.shape
        kv = context_vt = None
        if isinstance(context, ContextProjections):
            kv = context.kv.get(id(self)) if self.lora is None else None
            context_vt = context.vt
            context = context.context
        ctx = x if context is None else context
        if kv is not None:
            q, (k, v) = _lin(self.to_q, x), kv
            vt = context_vt.get(id(self)) if context_vt is not None else None
            if vt is not None and N >= 256 and _CTX_VT and q.shape[-1] == 64 * self.heads and not torch.is_grad_enabled():
                q4, k4 = q.view(B, N, self.heads, 64), k.view(B, k.shape[1], self.heads, 64)
                if nn_ops._attention_d64_layout_ok(q4, k4, k4):
                    return _lin(self.to_out[0], nn_ops.attention_d64_vt_strided(q4, k4, vt, k.shape[1]))
        elif _FUSED_QKV and context is None and self.lora is None and x.is_cuda and self.to_q.bias is None and \
                not self.to_q.weight.requires_grad and not torch.is_grad_enabled():
            # frozen self-attention: ONE [C, 2C] projection for q | k (the attention kernel reads the two strided views)
            # and V TRANSPOSED straight from a GEMM, W_v x^T = [B, C, N] -- the layout the kernel's P V product consumes
            # (round 3: its K fragment rows are permuted instead of V^T's keys), so no transposing pre-pass runs
            src = (self.to_q.weight.data_ptr(), self.to_q.weight._version, self.to_k.weight._version,
                   self.to_v.weight._version, x.dtype)
            if getattr(self, "_wqkv_src", None) != src:
                self._wqk = torch.cat([self.to_q.weight, self.to_k.weight], dim=0).detach().contiguous()
                self._wqkv_src = src
            q, k = (linear_320(x, self._wqk) if _LIN320 and linear_320_supported(x, self._wqk)
                    else _lib_linear(x, self._wqk)).chunk(2, dim=-1)
            q = q.view(B, N, self.heads, -1)
            k = k.view(B, N, self.heads, -1)
            if _VT_GEMM and N % 64 == 0 and N >= 256 and q.shape[-1] == 64 and x.dtype == torch.bfloat16:
                vt = torch.matmul(self.to_v.weight.detach(), x.transpose(1, 2))      # [B, C, N]
                return _lin(self.to_out[0], attention_d64_vt(q, k, vt))
            v = _lin(self.to_v, x)
            q, k = q.reshape(B, N, -1), k.reshape(B, N, -1)
        elif self.lora is not None:
            q = self._lora_lin(self.to_q, x, "to_q_lora")
            k = self._lora_lin(self.to_k, ctx, "to_k_lora")
            v = self._lora_lin(self.to_v, ctx, "to_v_lora")
        else:
            q, k, v = _lin(self.to_q, x), _lin(self.to_k, ctx), _lin(self.to_v, ctx)
        q = q.view(B, N, self.heads, -1)
        k = k.view(B, ctx.shape[1], self.heads, -1)
        v = v.view(B, ctx.shape[1], self.heads, -1)
        if N >= 256 and q.shape[-1] == 64 and nn_ops.attention_d64_train_supported(q, k, v):
            # training pass (LoRA UNet): own forward kernel + the library's flash backward on its log-sum-exp
            o = nn_ops.attention_d64_train(q, k, v)
        elif N >= 256 and q.shape[-1] == 64 and attention_d64_supported(q, k, v):
            # spatial self-attention and (round 2) cross-attention over the 77 text tokens, head_dim 64, inference: own fused kernel (nn_ops.attention_d64; 1.1-1.3x SDPA at
            # batch 16, 1.8-2x at batch 2 on MI355X, tools/attn_bench.py)
            o = attention_d64(q, k, v)
        else:
            o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
            o = o.transpose(1, 2).reshape(B, N, -1)
        if self.lora is not None:
            return self._lora_lin(self.to_out[0], o, "to_out_lora")
        return _lin(self.to_out[0], o)

    def _lora_lin(self, mod: nn.Linear, x, name):
        """``mod(x) + lora_scale * up(down(x))`` (LoRAAttnProcessor, lora_unet.py:415-422).  Frozen projection, bf16 activations,
        fp32 adapters: ONE autograd node (nn_ops.lora_linear) -- the projection on its inference routing, the adapter branch
        added by one launch, and a backward pass whose input gradient already holds both branches."""
        layer = self.lora[name]
        dw, uw = layer.down.weight, layer.up.weight
        w, b = mod.weight, mod.bias
        if _LORA_FUSED and _LORA_LINEAR and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and \
                not w.requires_grad and (b is None or not b.requires_grad) and dw.dtype == torch.float32 and \
                uw.dtype == torch.float32 and dw.shape[0] == 4 and uw.shape[1] == 4 and x.shape[-1] % 8 == 0 and \
                w.shape[0] % 8 == 0 and dw.is_contiguous() and uw.is_contiguous():
            return nn_ops.lora_linear(x, w, dw, uw, self.lora_scale, lambda t: _lin(mod, t))
        return self._lora_add(_lin(mod, x), x, name)

    def _lora_add(self, base, x, name):
        """``base + lora_scale * up(down(x))`` (LoRAAttnProcessor): own fused rank-4 kernels, forward and backward, for bf16
        activations with fp32 adapters (nn_ops.lora_branch: one launch instead of cast, GEMM, GEMM, cast, scale, add)."""
        layer = self.lora[name]
        dw, uw = layer.down.weight, layer.up.weight
        if _LORA_FUSED and nn_ops.lora_branch_supported(x, base, dw, uw):
            return nn_ops.lora_branch(x, base, dw, uw, self.lora_scale)
        return base + self.lora_scale * layer(x)


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        w = self.proj.weight
        if _LIN320 and x.is_cuda and not torch.is_grad_enabled() and not w.requires_grad and tuple(w.shape) == (2560, 320) and \
                linear_320_supported(x, w, self.proj.bias):
            return linear_320_geglu(x, w, self.proj.bias)      # projection + GEGLU in one kernel (64x64-token blocks)
        if _OWN_GEMM and x.is_cuda and not torch.is_grad_enabled() and not w.requires_grad and w.shape[1] != 320 and \
                (x.numel() // x.shape[-1]) * route_scale() >= _OWN_GEMM_MIN_ROWS and gemm_supported(x, w, self.proj.bias, geglu=True):
            # K = 640 / 1280 (the 32^2 / 16^2 / 8^2-token blocks): the own GEMM with the GEGLU as its epilogue -- 1.10-1.22x the
            # library GEMM + geglu_kernel pair (profiles/r06_gemm_shapes.txt), the [M][2 inner] projection output never
            # exists, and a row's bits do not depend on M, so batch-invariant selection needs no padded row set here (the
            # row threshold looks at the single-rank batch: every rank takes the same branch)
            return gemm_geglu(x, w, self.proj.bias)
        return geglu(_lin(self.proj, x))   # h * gelu(gate), fused on the GPU (nn_ops.geglu)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        return _lin(self.net[2], self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context, defer_ff_bias: bool = False):
        # x = x + attn1(norm1(x)); x = x + attn2(norm2(x), ctx); x = x + ff(norm3(x)) with each residual add fused
        # into the LayerNorm that follows it (nn_ops.add_layer_norm; plain torch ops when gradients are needed)
        _, n = add_layer_norm(x, None, self.norm1)
        x, n = add_layer_norm(x, self.attn1(n), self.norm2)
        x, n = add_layer_norm(x, self.attn2(n, context), self.norm3)
        if defer_ff_bias:
            # last residual inside the GEMM (beta = 1): x + gelu-gated(n) @ W2^T; the caller owes the constant b2
            g = self.ff.net[0](n)
            return _lib_addmm(x.reshape(-1, x.shape[-1]), g.reshape(-1, g.shape[-1]), self.ff.net[2].weight.t()).view_as(x)
        return x + self.ff(n)


class Transformer2DModel(nn.Module):
    def __init__(self, channels: int, heads: int, dim_head: int, cross_dim: int, groups: int = 32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Linear(channels, channels)  # use_linear_projection
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, dim_head, cross_dim)])
        self.proj_out = nn.Linear(channels, channels)

    def _folded_out_bias(self, dtype):
        w, b2 = self.proj_out.weight, self.transformer_blocks[0].ff.net[2].bias
        key = (w.data_ptr(), w._version, b2._version, self.proj_out.bias._version, dtype)
        if getattr(self, "_fold_key", None) != key:
            with torch.no_grad():
                self._fold_bias = (self.proj_out.bias.float() + w.float() @ b2.float()).to(dtype)
            self._fold_key = key
        return self._fold_bias

    def forward(self, x, context):
        B, C, H, W = x.shape
        h = _gn(self.norm, x, False)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)  # free for channels_last activations
        h = _lin(self.proj_in, h)
        if len(self.transformer_blocks) == 1 and h.is_cuda and not torch.is_grad_enabled() and \
                not self.proj_out.weight.requires_grad:
            # frozen inference: the block's feed-forward output bias b2 is a constant added right before proj_out,
            # so it moves into proj_out's bias (W_out b2 + b_out) and the residual add moves into the GEMM
            h = self.transformer_blocks[0](h, context, defer_ff_bias=True)
            h = _lin(self.proj_out, h, self._folded_out_bias(h.dtype))
        else:
            for blk in self.transformer_blocks:
                h = blk(h, context)
            h = self.proj_out(h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + x


class Downsample2D(nn.Module):
    def __init__(self, ch: int, padding: int = 1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if conv3x3_s2_supported(x, self.conv.weight):
            # stride-2 MFMA kernel; the VAE encoder's asymmetric pad (0,1,0,1) is part of its tap geometry
            return conv3x3_s2(x, self.conv.weight, self.conv.bias, self.padding)
        nn_ops._note_fallback("sd21.Downsample2D", x, "needs frozen bf16 weights with Cin % 64 == 0 and Cout % 64 == 0")
        if self.padding == 0:  # VAE encoder: asymmetric pad (0,1,0,1)
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        if upsample2x_conv3x3_supported(x, self.conv.weight) and not self.conv.bias.requires_grad:
            return upsample2x_conv3x3(x, self.conv.weight, self.conv.bias)   # no upsampled tensor, 16 taps not 36
        return _conv3(self.conv, F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _DownBlock(nn.Module):
    def __init__(self, in_ch, out_ch, temb_ch, n_layers, heads, cross_dim, attn: bool, down: bool):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, temb_ch) for i in range(n_layers)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(out_ch, heads, out_ch // heads, cross_dim) for _ in range(n_layers)]) if attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if down else None

    def forward(self, x, temb, context, skips):
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, context)
            skips.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            skips.append(x)
        return x


class _UpBlock(nn.Module):
    def __init__(self, in_ch, prev_ch, out_ch, temb_ch, n_layers, heads, cross_dim, attn: bool, up: bool):
        super().__init__()
        res = []
        for i in range(n_layers):
            skip_ch = in_ch if i == n_layers - 1 else out_ch
            res_in = prev_ch if i == 0 else out_ch
            res.append(ResnetBlock2D(res_in + skip_ch, out_ch, temb_ch))
        self.resnets = nn.ModuleList(res)
        self.attentions = nn.ModuleList(
            [Transformer2DModel(out_ch, heads, out_ch // heads, cross_dim) for _ in range(n_layers)]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if up else None

    def forward(self, x, temb, context, skips):
        for i, res in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = res(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, context)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _MidBlockCrossAttn(nn.Module):
    def __init__(self, ch, temb_ch, heads, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb_ch), ResnetBlock2D(ch, ch, temb_ch)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, ch // heads, cross_dim)])

    def forward(self, x, temb, context):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, context)
        return self.resnets[1](x, temb)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.linear_1 = nn.Linear(in_ch, out_ch)
        self.linear_2 = nn.Linear(out_ch, out_ch)

    def forward(self, x):
        return _lin(self.linear_2, F.silu(_lin(self.linear_1, x)))


def sinusoidal_timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`` -> [B, dim] fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class UNet2DConditionModel(nn.Module):
    """SD-2.1 denoiser.  ``forward(sample[B,4,h,w], timestep[B], encoder_hidden_states[B,77,1024])``
    returns the predicted noise ``[B,4,h,w]`` (the reference reads ``.sample`` of diffusers' output
    object: stable_diffusion_guidance.py:153-157)."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20)):
        super().__init__()
        ch = block_out_channels
        temb_ch = ch[0] * 4
        self.block_out_channels = tuple(ch)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb_ch)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, c in enumerate(ch):
            inp, out = out, c
            last = i == len(ch) - 1
            self.down_blocks.append(_DownBlock(inp, out, temb_ch, layers_per_block, attention_head_dim[i],
                                               cross_attention_dim, attn=not last, down=not last))
        self.mid_block = _MidBlockCrossAttn(ch[-1], temb_ch, attention_head_dim[-1], cross_attention_dim)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        rev_heads = list(reversed(attention_head_dim))
        out = rev[0]
        for i, c in enumerate(rev):
            prev, out = out, c
            inp = rev[min(i + 1, len(ch) - 1)]
            last = i == len(ch) - 1
            self.up_blocks.append(_UpBlock(inp, prev, out, temb_ch, layers_per_block + 1, rev_heads[i],
                                           cross_attention_dim, attn=i != 0, up=not last))
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)

    def extra_embedding(self, batch: int, **kwargs):
        """Hook for subclasses: an additive term for the time embedding (None here)."""
        return None

    def _project_context(self, ctx):
        """Every cross-attention's K and V of the text embeddings in one GEMM (``ContextProjections``) for a frozen
        UNet without gradients; otherwise the embeddings themselves."""
        if not ctx.is_cuda or torch.is_grad_enabled():
            return ctx
        atts = [m.attn2 for m in self.modules() if isinstance(m, BasicTransformerBlock)]
        if not atts or any(a.to_k.weight.requires_grad or a.to_k.bias is not None or a.lora is not None for a in atts):
            return ctx
        first = atts[0].to_k.weight
        key = (first.data_ptr(), ctx.dtype, len(atts)) + tuple(t._version for a in atts for t in (a.to_k.weight, a.to_v.weight))
        cache = getattr(self, "_ctx_cat", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = torch.cat([t for a in atts for t in (a.to_k.weight, a.to_v.weight)], dim=0).to(ctx.dtype).contiguous()
                wv = torch.cat([a.to_v.weight for a in atts], dim=0).to(ctx.dtype).contiguous()
            cache = self._ctx_cat = (key, w, wv)
        with torch.no_grad():
            allp = _lib_linear(ctx, cache[1])
            vt_all = None
            if _CTX_VT and ctx.dtype == torch.bfloat16 and all(a.to_v.out_features == 64 * a.heads for a in atts):
                # V^T of every layer: W_v_cat [sum C, 1024] x context^T, the context zero-padded to a whole number of 64-key
                # tiles (padded keys: V^T columns = 0, what the attention kernel expects); one batched GEMM per UNet call
                T = ctx.shape[1]
                Tp = (T + 63) // 64 * 64
                ctx_p = F.pad(ctx, (0, 0, 0, Tp - T)) if Tp != T else ctx
                vt_all = torch.matmul(cache[2], ctx_p.transpose(1, 2))          # [B, sum C, Tp]
        kv, vt, off, voff = {}, {}, 0, 0
        for a in atts:
            c = a.to_k.out_features
            kv[id(a)] = (allp[..., off:off + c], allp[..., off + c:off + 2 * c])
            if vt_all is not None:
                vt[id(a)] = vt_all[:, voff:voff + c]
            off += 2 * c
            voff += c
        return ContextProjections(ctx, kv, vt)

    def _project_temb(self, temb):
        """All blocks' per-image conv1 biases in one GEMM (``TembProjections``).  The projection weights must be frozen; the
        time embedding itself may need a gradient (the LoRA UNet's camera / shading embedding is part of it): the GEMM and the
        SiLU before it then run under autograd ONCE, the blocks take ``torch.split`` views of the result, and the backward pass
        is one concatenation of the 22 bias gradients, one GEMM and one SiLU backward instead of 22 of each plus 21
        accumulations into the embedding's gradient (round 5; ``GD_TEMB_TRAIN_CAT=0`` = per block as before)."""
        train = torch.is_grad_enabled() and temb.requires_grad
        if not temb.is_cuda or (train and not _TEMB_TRAIN_CAT):
            return temb
        blocks = [m for m in self.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]
        if not blocks or any(b.time_emb_proj.weight.requires_grad or b.conv1.bias.requires_grad for b in blocks):
            return temb
        first = blocks[0].time_emb_proj.weight
        key = (first.data_ptr(), temb.dtype, len(blocks)) + tuple(
            t._version for b in blocks for t in (b.time_emb_proj.weight, b.time_emb_proj.bias, b.conv1.bias))
        cache = getattr(self, "_temb_cat", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = torch.cat([b.time_emb_proj.weight for b in blocks], dim=0).to(temb.dtype).contiguous()
                bias = torch.cat([b.time_emb_proj.bias + b.conv1.bias for b in blocks], dim=0).to(temb.dtype)
            cache = self._temb_cat = (key, w, bias)
        if train:
            parts = torch.split(_lib_linear(F.silu(temb), cache[1], cache[2]), [b.time_emb_proj.out_features for b in blocks], dim=1)
            return TembProjections(temb, {id(b): p for b, p in zip(blocks, parts)})
        with torch.no_grad():
            allp = _lib_linear(F.silu(temb), cache[1], cache[2])
        out, off = {}, 0
        for b in blocks:
            c = b.time_emb_proj.out_features
            out[id(b)] = allp[:, off:off + c]
            off += c
        return TembProjections(temb, out)

    fp8 = None    # nn_ops.Fp8State: e4m3 convolutions in the no-grad forward (enable_fp8); None = bf16 everywhere

    def enable_fp8(self, state=None):
        """Run the 3x3 convolutions of the no-grad forward in e4m3 (nn_ops.Fp8State).  The first no-grad forward(s)
        calibrate the activation ranges in bf16 until ``fp8.mode`` is set to "run" (the guidance does that after one
        call).  Returns the state."""
        from ..nn_ops import Fp8State
        self.fp8 = state if state is not None else Fp8State()
        return self.fp8

    def forward(self, sample, timestep, encoder_hidden_states, **kwargs):
        if self.fp8 is not None and not torch.is_grad_enabled() and sample.is_cuda:
            _FP8_ACTIVE[0] = self.fp8
            try:
                return self._forward(sample, timestep, encoder_hidden_states, **kwargs)
            finally:
                _FP8_ACTIVE[0] = None
        return self._forward(sample, timestep, encoder_hidden_states, **kwargs)

    def _forward(self, sample, timestep, encoder_hidden_states, **kwargs):
        dtype = self.conv_in.weight.dtype
        if timestep.dim() == 0:
            timestep = timestep[None].expand(sample.shape[0])
        temb = self.time_embedding(sinusoidal_timestep_embedding(timestep, self.block_out_channels[0]).to(dtype))
        extra = self.extra_embedding(sample.shape[0], **kwargs)
        if extra is not None:
            temb = temb + extra.to(dtype)
        temb = self._project_temb(temb)
        x = conv3x3_small_cin(sample.to(dtype).contiguous(memory_format=torch.channels_last), self.conv_in.weight,
                              self.conv_in.bias)
        ctx = self._project_context(encoder_hidden_states.to(dtype))
        skips = [x]
        for blk in self.down_blocks:
            x = blk(x, temb, ctx, skips)
        x = self.mid_block(x, temb, ctx)
        for blk in self.up_blocks:
            x = blk(x, temb, ctx, skips)
        return _conv3(self.conv_out, _gn(self.conv_norm_out, x, True))


class LoraUNet2DConditionModel(UNet2DConditionModel):
    """The NeTF stage's trainable "q" network: an SD-2.1 UNet with LoRA adapters on every attention
    (rank 4) plus a camera-pose MLP and per-shading embeddings added to the time embedding
    (Garment_Deformer_NeTF/netf/vsd/lora_unet.py:415-422,632-645; adapters installed at
    Garment_Deformer_NeTF/netf/trainer.py:88-101).  ``forward(x, t, text, c=pose[B,16], shading=...)``.
    Base weights are frozen; only the adapters, ``camera_emb`` and the shading embeddings train."""

    def __init__(self, rank: int = 4, **kw):
        super().__init__(**kw)
        temb_ch = self.block_out_channels[0] * 4
        self.camera_emb = nn.Sequential(nn.Linear(16, temb_ch), nn.SiLU(), nn.Linear(temb_ch, temb_ch))
        self.lambertian_emb = nn.Parameter(torch.randn(1, temb_ch))
        self.textureless_emb = nn.Parameter(torch.randn(1, temb_ch))
        self.normal_emb = nn.Parameter(torch.randn(1, temb_ch))
        self.lora_layers = nn.ModuleList()
        for m in self.modules():
            if isinstance(m, Attention):
                self.lora_layers.append(m.add_lora(rank))

    def adapters_to_fp32(self):
        """Keep the trainable rank-4 adapters (and so the whole LoRA branch: ``LoRALinearLayer.forward`` computes in its
        weights' dtype) in fp32 while the frozen base runs bf16.  The reference trains them in fp32
        (sd_vsd_utils.py:35); an adapter gradient is a sum over all tokens that cancels to ~1e-3 of its terms, so bf16
        intermediates (the rank-4 activations, the gradient GEMM's output) leave mostly rounding noise in it.  The
        branch is [tokens, C] x [C, 4]: its fp32 cost is negligible."""
        self.lora_layers.float()
        return self

    def trainables_to_fp32(self):
        """Adapters AND the camera MLP / shading embeddings in fp32 (the reference trains every one of them in fp32,
        sd_vsd_utils.py:35 / trainer.py:129-137); the frozen base stays bf16.  ``extra_embedding`` computes in the MLP's dtype
        and casts the [B, 1280] result to the time embedding's."""
        self.adapters_to_fp32()
        self.camera_emb.float()
        for p in (self.lambertian_emb, self.textureless_emb, self.normal_emb):
            p.data = p.data.float()
        return self

    def freeze_base(self):
        for p in self.parameters():
            p.requires_grad_(False)
        train = list(self.lora_layers.parameters()) + list(self.camera_emb.parameters()) + \
            [self.lambertian_emb, self.textureless_emb, self.normal_emb]
        for p in train:
            p.requires_grad_(True)
        return train

    def forward(self, sample, timestep, encoder_hidden_states, **kwargs):
        # training pass: the weight gradients of all adapted projections leave through ONE grouped launch per stage at the end
        # of the backward pass (nn_ops.LoraGradGroup) instead of two ~5 us launches per projection
        if torch.is_grad_enabled() and sample.is_cuda:
            with nn_ops.lora_grad_group():
                return super().forward(sample, timestep, encoder_hidden_states, **kwargs)
        return super().forward(sample, timestep, encoder_hidden_states, **kwargs)

    def extra_embedding(self, batch: int, c=None, shading: str = "albedo"):
        dev, dt = self.camera_emb[0].weight.device, self.camera_emb[0].weight.dtype
        if c is None:
            c = torch.zeros(batch, 16, device=dev)
        emb = self.camera_emb(c.to(device=dev, dtype=dt))
        if shading == "textureless":
            emb = emb + self.textureless_emb
        elif shading == "lambertian":
            emb = emb + self.lambertian_emb
        elif shading == "normal":
            emb = emb + self.normal_emb
        else:
            assert shading == "albedo"
        return emb


# ----------------------------------------------------------------------------------------------
# VAE encoder
# ----------------------------------------------------------------------------------------------


class _VAEAttention(nn.Module):
    """Single-head spatial self-attention of the VAE mid block (N = h*w tokens, d = channels)."""

    def __init__(self, ch: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch)])

    def _qkv_scaled(self, dtype, scale):
        ws = (self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_q.bias, self.to_k.bias, self.to_v.bias)
        key = (ws[0].data_ptr(), dtype) + tuple(t._version for t in ws)
        if getattr(self, "_qkv_key", None) != key:
            with torch.no_grad():
                w = torch.cat([ws[0].float() * scale, ws[1].float(), ws[2].float()], dim=0).to(dtype).contiguous()
                b = torch.cat([ws[3].float() * scale, ws[4].float(), ws[5].float()], dim=0).to(dtype)
            self._qkv_cache, self._qkv_key = (w, b), key
        return self._qkv_cache

    def forward(self, x):
        B, C, H, W = x.shape
        h = _gn(self.group_norm, x, False).permute(0, 2, 3, 1).reshape(B, H * W, C)
        if x.is_cuda:
            # one head of dim 512: plain GEMMs + a softmax (hipBLASLt) beat the fused kernel, whose backward for
            # head_dim 512 runs at ~180 TFLOP/s; the [B,N,N] bf16 score matrix (34 MB per image at N = 4096) is
            # cheap on a 288 GB part.  Frozen weights: one [C, 3C] projection with the softmax scale folded into its
            # q rows (no N x C scaling pass forward or backward); the bmm's read the strided q / k / v views.
            o = None
            if _FUSED_QKV and not (self.to_q.weight.requires_grad or self.to_k.weight.requires_grad or self.to_v.weight.requires_grad):
                qkv = _lib_linear(h, *self._qkv_scaled(h.dtype, C ** -0.5))
                if _VAE_ATTN_NODE and nn_ops.single_head_attention_supported(qkv):
                    # scores, own row softmax (forward and backward, in place) and the three GEMM gradients written into one
                    # [B, N, 3C] tensor: one autograd node (nn_ops._SingleHeadAttention)
                    o = nn_ops.single_head_attention(qkv)
                else:
                    q, k, v = qkv.chunk(3, dim=-1)
            else:
                q, k, v = _lin(self.to_q, h) * (C ** -0.5), _lin(self.to_k, h), _lin(self.to_v, h)
            if o is None:
                p = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
                o = torch.bmm(p, v)
        else:
            q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
            o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = _lin(self.to_out[0], o).reshape(B, H, W, C).permute(0, 3, 1, 2)
        return x + o


class _VAEDownBlock(nn.Module):
    def __init__(self, in_ch, out_ch, n_layers, down: bool):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, None, eps=1e-6) for i in range(n_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch, padding=0)]) if down else None

    def forward(self, x, next_norm=None):
        for i, r in enumerate(self.resnets):   # the next block's GroupNorm gets its statistics from this block's conv2
            last = i + 1 == len(self.resnets)
            x = r(x, next_norm=(next_norm if self.downsamplers is None else None) if last else self.resnets[i + 1].norm1)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class _VAEMidBlock(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, eps=1e-6), ResnetBlock2D(ch, ch, None, eps=1e-6)])
        self.attentions = nn.ModuleList([_VAEAttention(ch)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, in_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4):
        super().__init__()
        ch = block_out_channels
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = ch[0]
        for i, c in enumerate(ch):
            inp, out = out, c
            self.down_blocks.append(_VAEDownBlock(inp, out, layers_per_block, down=i != len(ch) - 1))
        self.mid_block = _VAEMidBlock(ch[-1])
        self.conv_norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = conv3x3_small_cin(x.contiguous(memory_format=torch.channels_last), self.conv_in.weight, self.conv_in.bias,
                              next_norm=self.down_blocks[0].resnets[0].norm1)
        for i, b in enumerate(self.down_blocks):
            x = b(x, next_norm=self.mid_block.resnets[0].norm1 if i + 1 == len(self.down_blocks) else None)
        x = self.mid_block(x)
        return _conv3(self.conv_out, _gn(self.conv_norm_out, x, True))


class DiagonalGaussianDistribution:
    def __init__(self, parameters: torch.Tensor):
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, noise: Optional[torch.Tensor] = None, generator=None) -> torch.Tensor:
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _EncodeOutput:
    def __init__(self, dist):
        self.latent_dist = dist


class _VAEConfig:
    scaling_factor = 0.18215


class _QuantConv(nn.Conv2d):
    """``AutoencoderKL.quant_conv`` = ``nn.Conv2d(8, 8, 1)`` (same parameters and state_dict keys): frozen bf16 weights on the
    GPU run on the own one-vector-per-pixel kernel, forward and input gradient (nn_ops.conv1x1_c8) -- MIOpen chose its naive
    fp64-accumulating kernel for this layer, the last library convolution of the step; anything else is ``F.conv2d``."""

    def forward(self, x):
        if nn_ops.conv1x1_c8_supported(x, self.weight, self.bias):
            return nn_ops.conv1x1_c8(x, self.weight, self.bias)
        nn_ops._note_fallback("sd21._QuantConv", x, "needs frozen contiguous bf16 8 -> 8 weights and a 16-byte aligned tensor")
        return super().forward(x)


class AutoencoderKLEncoder(nn.Module):
    """Encoder half of AutoencoderKL: ``encode(x).latent_dist.sample()`` as the reference calls it
    (stable_diffusion_guidance.py:165-166).  The decoder is only used by ``guidance_eval`` previews
    (disabled in the pipeline: GaussianDreamer.py:244) and is out of scope."""

    config = _VAEConfig()

    def __init__(self, block_out_channels=(128, 256, 512, 512)):
        super().__init__()
        self.encoder = Encoder(block_out_channels=block_out_channels)
        self.quant_conv = _QuantConv(8, 8, 1)

    def encode(self, x):
        return _EncodeOutput(DiagonalGaussianDistribution(self.quant_conv(self.encoder(x.to(self.quant_conv.weight.dtype)))))


# ----------------------------------------------------------------------------------------------
# scheduler
# ----------------------------------------------------------------------------------------------


class _SchedulerConfig:
    num_train_timesteps = 1000
    prediction_type = "epsilon"


class DDIMScheduler:
    """The subset the guidance uses: ``alphas_cumprod``, ``config.num_train_timesteps``, ``add_noise``
    (stable_diffusion_guidance.py:129-131,238) and, for the ``guidance_eval`` previews (:505-579), ``set_timesteps`` /
    ``timesteps`` / ``step`` -- the published DDIM update (Song et al. 2021, eq. 12) with the SD-2.1 scheduler config
    (scaled_linear betas, 1000 steps, "leading" spacing, ``steps_offset`` 1, ``set_alpha_to_one`` False, no sample
    clipping).  diffusers is absent: scheduler numerics are restated, not pinned."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, num_train_timesteps=1000, prediction_type="epsilon",
                 steps_offset=1):
        self.config = _SchedulerConfig()
        self.config.num_train_timesteps = num_train_timesteps
        self.config.prediction_type = prediction_type
        self.config.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]         # set_alpha_to_one = False
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.long)

    def set_timesteps(self, num_inference_steps: int, device=None):
        """"leading" spacing: multiples of T / n, descending, shifted by ``steps_offset``."""
        self.num_inference_steps = int(num_inference_steps)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        ts = (torch.arange(0, self.num_inference_steps, dtype=torch.long) * ratio).flip(0) + self.config.steps_offset
        self.timesteps = ts.to(device) if device is not None else ts

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, variance_noise=None):
        """One reverse step x_t -> x_{t - T/n}: ``{"prev_sample", "pred_original_sample"}``.  ``eta`` scales the DDIM
        variance sigma_t^2 = (1 - abar_prev) / (1 - abar_t) (1 - abar_t / abar_prev); eta = 1 is the DDPM-like sampler."""
        if self.num_inference_steps is None:
            raise ValueError("DDIMScheduler.step: call set_timesteps first")
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        ac = self.alphas_cumprod.to(device=sample.device, dtype=torch.float32)
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else self.final_alpha_cumprod.to(sample.device)
        b_t = 1 - a_t
        if self.config.prediction_type == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.config.prediction_type == "v_prediction":
            x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise ValueError(f"prediction_type {self.config.prediction_type}")
        variance = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        prev = a_prev ** 0.5 * x0 + (1 - a_prev - std ** 2) ** 0.5 * eps
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, device=model_output.device,
                                             dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return {"prev_sample": prev.to(sample.dtype), "pred_original_sample": x0.to(sample.dtype)}

    def _alphas_on(self, device, dtype):
        """``alphas_cumprod`` on ``device`` in ``dtype``, copied there ONCE: the table lives on the host (diffusers keeps it there),
        and a pageable host-to-device copy per call waits for everything queued on the stream -- three full synchronisations per
        VSD iteration before round 5 (tools/vsd_host_timeline.py)."""
        if not _PINNED_TABLES:
            return self.alphas_cumprod.to(device=device, dtype=dtype)
        key = (str(device), dtype)
        hit = self._alphas_cache.get(key) if hasattr(self, "_alphas_cache") else None
        if hit is None or hit[0] is not self.alphas_cumprod:
            if not hasattr(self, "_alphas_cache"):
                self._alphas_cache = {}
            hit = self._alphas_cache[key] = (self.alphas_cumprod, self.alphas_cumprod.to(device=device, dtype=dtype))
        return hit[1]

    def add_noise(self, original_samples, noise, timesteps):
        ac = self._alphas_on(original_samples.device, original_samples.dtype)
        sqrt_a = ac[timesteps] ** 0.5
        sqrt_1ma = (1 - ac[timesteps]) ** 0.5
        while sqrt_a.dim() < original_samples.dim():
            sqrt_a = sqrt_a.unsqueeze(-1)
            sqrt_1ma = sqrt_1ma.unsqueeze(-1)
        return sqrt_a * original_samples + sqrt_1ma * noise

    def get_velocity(self, sample, noise, timesteps):
        """v-prediction target: sqrt(abar) * eps - sqrt(1 - abar) * x0."""
        ac = self._alphas_on(sample.device, sample.dtype)
        sqrt_a = ac[timesteps] ** 0.5
        sqrt_1ma = (1 - ac[timesteps]) ** 0.5
        while sqrt_a.dim() < sample.dim():
            sqrt_a = sqrt_a.unsqueeze(-1)
            sqrt_1ma = sqrt_1ma.unsqueeze(-1)
        return sqrt_a * noise - sqrt_1ma * sample


# ----------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------


def init_random_(module: nn.Module, seed: int = 0) -> nn.Module:
    """Deterministic random init ON THE MODULE'S DEVICE (fast for the 866 M-parameter UNet) at a
    scale that keeps activations O(1) through ~60 residual blocks: N(0, 1/fan_in) weights,
    residual-branch output layers scaled down, zero biases, unit norm gains."""
    gens = {}
    with torch.no_grad():
        for name, p in module.named_parameters():
            if "lora" in name:   # adapters keep their constructor init (down ~ N(0, 1/rank), up = 0)
                continue
            g = gens.get(p.device)
            if g is None:
                g = gens[p.device] = torch.Generator(device=p.device).manual_seed(seed)
            if p.dim() > 1:
                fan_in = p[0].numel()
                p.normal_(0.0, 1.0 / math.sqrt(fan_in), generator=g)
                if name.endswith(("conv2.weight", "to_out.0.weight", "net.2.weight", "proj_out.weight")):
                    p.mul_(0.2)
            elif name.endswith(".bias"):
                p.zero_()
            else:
                p.fill_(1.0)
    return module


def load_diffusers_weights(module: nn.Module, path: str, prefix: str = "") -> None:
    """Load a diffusers ``*.safetensors`` shard (UNet or VAE) by key; names already match."""
    from safetensors.torch import load_file
    sd = load_file(path)
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    own = module.state_dict()
    missing = [k for k in own if k not in sd]
    if missing:
        raise KeyError(f"{len(missing)} parameters missing in {path}, e.g. {missing[:3]}")
    module.load_state_dict({k: sd[k] for k in own})
