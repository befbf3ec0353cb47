"""Torch-facing wrappers of ``libgd_nn.so`` (C-ABI: include/gd_nn.h) -- hand-written gfx950 kernels for
the guidance step.  On HIP tensors the kernels are mandatory (missing library -> error, no silent
eager fallback); on CPU tensors (tests, cpu_baseline) the plain PyTorch ops of the same math run.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgd_nn.so")
_LIB_OVERRIDDEN = False
_lib = None


def use_library(path: str) -> None:
    """Point the wrappers at another build of libgd_nn.so BEFORE its first use -- same-box A/B timing of experimental
    builds by the scripts under tools/ (tools/ablib.py) and ``bench.py --nn-lib``; the package itself reads no environment
    variable for this.  Symbols an older build lacks are skipped at load (calling them raises)."""
    global _LIB_PATH, _LIB_OVERRIDDEN
    if _lib is not None:
        raise RuntimeError("libgd_nn.so is already loaded")
    _LIB_PATH, _LIB_OVERRIDDEN = os.path.abspath(path), True


_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "gd_nn_groupnorm_silu_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "gd_nn_groupnorm_silu_fused_supported": (_i, [_i, _i, _i, _i]),
    "gd_nn_groupnorm_silu_fused_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i]),
    "gd_nn_groupnorm_silu_fused_forward_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "gd_nn_groupnorm_silu_fused_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "gd_nn_groupnorm_silu_fused_backward_supported": (_i, [_i, _i, _i, _i]),
    "gd_nn_groupnorm_silu_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "gd_nn_groupnorm_ws_bytes": (C.c_size_t, [_i, _i]),
    "gd_nn_conv3x3_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_ws_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_forward_ws": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_size_t]),
    "gd_nn_conv_force_split": (_i, [_i]),
    "gd_nn_conv_set_route_scale": (_i, [_i]),
    "gd_nn_conv3x3_flip_weights": (_i, [_vp, _vp, _vp, _i, _i]),
    "gd_nn_conv3x3_first_dgrad_supported": (_i, [_i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_first_dgrad_weights": (_i, [_vp, _vp, _vp, _i, _i]),
    "gd_nn_conv3x3_first_dgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "gd_nn_groupnorm_stats": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "gd_nn_groupnorm_silu_forward_fp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _f]),
    "gd_nn_conv3x3_gn_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i]),
    "gd_nn_linear_320_supported": (_i, [C.c_int64, _i, _i]),
    "gd_nn_linear_320_forward": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64]),
    "gd_nn_linear_k320_forward": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i]),
    "gd_nn_linear_k320_geglu_forward": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i]),
    "gd_nn_linear_320_last_error": (C.c_char_p, []),
    "gd_nn_conv3x3_stat_rows": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_first_stat_rows": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_first_forward_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gd_nn_conv3x3_gn_forward_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i,
                                            _vp]),
    "gd_nn_conv3x3_forward_stats": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gd_nn_conv3x3_wino_supported": (_i, [_i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_wino_weights": (_i, [_vp, _vp, _vp, _i, _i]),
    "gd_nn_conv3x3_wino_weights_bytes": (C.c_size_t, [_i, _i]),
    "gd_nn_conv3x3_wino_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gd_nn_conv3x3_wino_gn_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gd_nn_conv3x3_wide_supported": (_i, [_i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_wide_weights": (_i, [_vp, _vp, _vp, _i, _i]),
    "gd_nn_conv3x3_wide_weights_bytes": (C.c_size_t, [_i, _i]),
    "gd_nn_conv3x3_wide_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gd_nn_conv3x3_wide_gn_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gd_nn_groupnorm_finish_partials": (_i, [_vp, _vp, _i, C.c_size_t, _i, _i, _i, _f, _vp]),
    "gd_nn_conv3x3_first_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_s2_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_s2_dgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_s2_ws_bytes": (C.c_size_t, [_i, _i, _i, _i, _i, _i, _i]),
    "gd_nn_conv3x3_s2_forward_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.c_size_t]),
    "gd_nn_conv3x3_s2_dgrad_ws": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.c_size_t]),
    "gd_nn_conv3x3_up2_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "gd_nn_conv_force_variant": (_i, [_i]),
    "gd_nn_linear_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i]),
    "gd_nn_gemm_supported": (_i, [C.c_int64, _i, _i, _i]),
    "gd_nn_gemm_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i]),
    "gd_nn_gemm_geglu_forward": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i]),
    "gd_nn_gemm_last_error": (C.c_char_p, []),
    "gd_nn_conv_profile_enable": (_i, [_i]),
    "gd_nn_conv_profile_reset": (_i, []),
    "gd_nn_conv_profile_read": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "gd_nn_conv_profile_read_bytes": (_i, [C.POINTER(C.c_double)]),
    "gd_nn_geglu_forward": (_i, [_vp, _vp, _vp, C.c_int64, _i]),
    "gd_nn_geglu_backward": (_i, [_vp, _vp, _vp, _vp, C.c_int64, _i]),
    "gd_nn_conv1x1_c8": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i]),
    "gd_nn_softmax_rows_forward": (_i, [_vp, _vp, _vp, C.c_int64, _i]),
    "gd_nn_softmax_rows_backward": (_i, [_vp, _vp, _vp, _vp, C.c_int64, _i]),
    "gd_nn_layernorm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, C.c_float]),
    "gd_nn_add_layernorm_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, C.c_int64, _i]),
    "gd_nn_attention_ws_bytes": (C.c_size_t, [_i, _i, _i]),
    "gd_nn_attention_d64_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_int64, _i, C.c_int64, _i,
                                         C.c_int64, _i, C.c_int64, _i, _f, _i]),
    "gd_nn_attention_d64_forward_lse": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_int64, _i, C.c_int64, _i,
                                             C.c_int64, _i, C.c_int64, _i, _f, _i]),
    "gd_nn_attention_bwd_ws_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "gd_nn_attention_d64_backward": (_i, [_vp] * 11 + [_i, _i, _i, _i] + [C.c_int64, _i] * 8 + [_f, _i]),
    "gd_nn_attention_d64_forward_vt": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_int64, _i, C.c_int64, _i, C.c_int64, _i,
                                            _f]),
    "gd_nn_attention_d64_forward_vt_strided": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_int64, _i, C.c_int64, _i, C.c_int64,
                                                    C.c_int64, _i, _f, _i]),
    "gd_nn_attention_last_error": (C.c_char_p, []),
    "gd_nn_vae_prologue_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "gd_nn_vae_prologue_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "gd_nn_sparsity_forward": (_i, [_vp, _vp, _vp, C.c_int64, _vp]),
    "gd_nn_sparsity_backward": (_i, [_vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "gd_nn_prologue_last_error": (C.c_char_p, []),
    "gd_nn_fp8_quantize": (_i, [_vp, _vp, _vp, C.c_int64, _f]),
    "gd_nn_fp8_pack_weights": (_i, [_vp, _vp, _vp, C.c_int64, _i, _i, _f]),
    "gd_nn_fp8_linear_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _i, _f]),
    "gd_nn_fp8_conv3x3_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f]),
    "gd_nn_fp8_last_error": (C.c_char_p, []),
    "gd_nn_conv_last_error": (C.c_char_p, []),
    "gd_nn_elementwise_last_error": (C.c_char_p, []),
    "gd_nn_lora_rowdot": (_i, [_vp, _vp, _vp, _vp, C.c_int64, _i, C.c_float, _i]),
    "gd_nn_lora_rank4_add": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i]),
    "gd_nn_lora_row_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, C.c_float, _i]),
    "gd_nn_lora_colreduce_scratch_floats": (C.c_size_t, [C.c_int64, _i]),
    "gd_nn_lora_colreduce": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _i, C.c_float, _i]),
    "gd_nn_lora_colreduce_pair_scratch_floats": (C.c_size_t, [C.c_int64, _i, _i]),
    "gd_nn_lora_colreduce_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i]),
    "gd_nn_lora_colreduce_pair_into": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _i]),
    "gd_nn_lora_colreduce_group_entry_bytes": (C.c_size_t, []),
    "gd_nn_lora_colreduce_group_desc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _i, C.POINTER(C.c_int)]),
    "gd_nn_lora_colreduce_group_launch": (_i, [_vp, _vp, _i, _i, _i, _i]),
    "gd_nn_lora_last_error": (C.c_char_p, []),
    "gd_nn_last_error": (C.c_char_p, []),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} not found: build with `python -m garmentdreamer_amd._build`; "
                               "the HIP guidance kernels have no fallback on GPU tensors")
        L = C.CDLL(_LIB_PATH)  # torch is imported above: one libamdhip64 per process
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if _LIB_OVERRIDDEN:      # an older experimental build under A/B (use_library): symbol not there yet
                    continue
                raise
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(ret, what, err="gd_nn_last_error"):
    """``err`` names the error-text getter of the translation unit ``what`` lives in (each csrc file keeps its own
    thread-local buffer: gd_nn_last_error is the GroupNorm file's)."""
    if ret < 0:
        raise RuntimeError(f"{what} failed ({ret}): {getattr(lib(), err)().decode()}")


_gn_ws_cache = {}
_WS_TAG = [None]


class workspace_tag:
    """Kernels launched (or CAPTURED) inside use a GroupNorm workspace of their own: two hipGraphs that may be replayed
    concurrently on different streams (sd_vsd.StableDiffusionVSD: the frozen UNet beside the LoRA UNet) were both captured on
    torch's one capture stream, so the stream alone would hand them the same accumulators."""

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        self.prev, _WS_TAG[0] = _WS_TAG[0], self.tag
        return self

    def __exit__(self, *exc):
        _WS_TAG[0] = self.prev
        return False


def _gn_workspace(x, N, groups):
    """The zero-initialised GroupNorm statistics workspace of (device, current stream, workspace_tag): every call leaves it zero
    (include/gd_nn.h), so it is allocated once and never memset again.  Keyed by stream because two streams may
    run GroupNorms concurrently; a workspace first needed during hipGraph capture is allocated (and kept alive
    here) in that graph's pool."""
    stream = torch.cuda.current_stream(x.device)
    key = (x.device.index, stream.cuda_stream, _WS_TAG[0])
    need = lib().gd_nn_groupnorm_ws_bytes(N, groups)
    ws = _gn_ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = _gn_ws_cache[key] = torch.zeros(max(need, 1 << 16), dtype=torch.uint8, device=x.device)
    return ws


def reset_workspaces():
    """Drop the cached workspaces (after an aborted hipGraph capture may have left one mid-update)."""
    _gn_ws_cache.clear()


# ---------------------------------------------------------------------------------------------
# library fallbacks: a bf16 GPU tensor that misses an own kernel's shape rules goes to PyTorch's op (MIOpen / aten).  That
# is what the fp32 reference runs of the tests need and what an odd shape deserves, but on the step's tensors it would put
# the library back into the iteration without anyone noticing -- so every such call is counted, and raises under strict mode.
# ---------------------------------------------------------------------------------------------
_FALLBACKS = {}
_STRICT = False


def set_strict_library(on: bool) -> None:
    """``True``: a bf16 GPU tensor falling back to a PyTorch library op raises instead of being counted."""
    global _STRICT
    _STRICT = bool(on)


def library_fallbacks(reset: bool = False) -> dict:
    """{"op: reason": calls} of the bf16-GPU calls that ran on a PyTorch library op since the last reset (bench.py puts the
    total in its line: 0 on the benchmark's shapes)."""
    out = dict(_FALLBACKS)
    if reset:
        _FALLBACKS.clear()
    return out


def _note_fallback(op: str, x, why: str) -> None:
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16):
        return          # CPU / fp32 reference runs: the PyTorch op IS the intended path
    key = f"{op}: {why}"
    if _STRICT:
        raise RuntimeError(f"nn_ops strict mode: {key} on a bf16 GPU tensor of shape {tuple(x.shape)} would run on the PyTorch library op")
    _FALLBACKS[key] = _FALLBACKS.get(key, 0) + 1


def _is_nhwc_bf16(x):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)


class _GroupNormSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu):
        N, Cc, H, W = x.shape
        L = lib()
        y = torch.empty_like(x, memory_format=torch.channels_last)
        ws = _gn_workspace(x, N, groups)
        mr = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
        w, b = weight.contiguous(), bias.contiguous()
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _check(L.gd_nn_groupnorm_silu_forward(stream, x.data_ptr(), y.data_ptr(), w.data_ptr(), b.data_ptr(), N,
                                                  H * W, Cc, groups, float(eps), int(silu), ws.data_ptr(),
                                                  mr.data_ptr()), "gd_nn_groupnorm_silu_forward")
        ctx.save_for_backward(x, w, b, mr)
        ctx.groups, ctx.silu = groups, silu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mr = ctx.saved_tensors
        N, Cc, H, W = x.shape
        L = lib()
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        ws = _gn_workspace(x, N, ctx.groups)
        sums = torch.empty(N * ctx.groups * 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _check(L.gd_nn_groupnorm_silu_backward(stream, x.data_ptr(), dy.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                   mr.data_ptr(), dx.data_ptr(), N, H * W, Cc, ctx.groups,
                                                   int(ctx.silu), ws.data_ptr(), sums.data_ptr(), None),
                   "gd_nn_groupnorm_silu_backward")
        return dx, None, None, None, None, None


# GD_NN_GN_SMALL=0: the two-pass GroupNorm also where the one-launch form applies (A/B timing in tools/; never set in tests)
_GN_FUSED_SMALL = os.environ.get("GD_NN_GN_SMALL", "1") != "0"


def _gn_fused_small(x, weight, bias, groups, eps, silu):
    """Inference GroupNorm(+SiLU) in one launch (gd_nn_groupnorm_silu_fused_forward): one workgroup per (image, group)."""
    N, Cc, H, W = x.shape
    y = torch.empty_like(x, memory_format=torch.channels_last)
    w, b = weight.contiguous(), bias.contiguous()
    with torch.cuda.device(x.device):
        _check(lib().gd_nn_groupnorm_silu_fused_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(),
                                                        y.data_ptr(), w.data_ptr(), b.data_ptr(), N, H * W, Cc, groups,
                                                        float(eps), int(silu)), "gd_nn_groupnorm_silu_fused_forward")
    return y


# GD_NN_GN_SMALL_TRAIN=0: the two-pass pair also in the training pass (A/B timing in tools/; never set in tests)
_GN_FUSED_TRAIN = os.environ.get("GD_NN_GN_SMALL_TRAIN", "1") != "0"


class _GroupNormSiLUSmall(torch.autograd.Function):
    """GroupNorm(+SiLU) WITH an input gradient on maps whose (image, group) slice fits a workgroup's registers (the LoRA
    UNet's training pass): one launch forward (statistics kept), one launch backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu):
        N, Cc, H, W = x.shape
        y = torch.empty_like(x, memory_format=torch.channels_last)
        mr = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
        w, b = weight.contiguous(), bias.contiguous()
        with torch.cuda.device(x.device):
            _check(lib().gd_nn_groupnorm_silu_fused_forward_stats(
                torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), y.data_ptr(), w.data_ptr(), b.data_ptr(), N,
                H * W, Cc, groups, float(eps), int(silu), mr.data_ptr()), "gd_nn_groupnorm_silu_fused_forward_stats")
        ctx.save_for_backward(x, w, b, mr)
        ctx.groups, ctx.silu = groups, silu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mr = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        L = lib()
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            if L.gd_nn_groupnorm_silu_fused_backward_supported(N, H * W, Cc, ctx.groups):
                _check(L.gd_nn_groupnorm_silu_fused_backward(stream, x.data_ptr(), dy.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                             mr.data_ptr(), dx.data_ptr(), N, H * W, Cc, ctx.groups,
                                                             int(ctx.silu)), "gd_nn_groupnorm_silu_fused_backward")
            else:       # x and dy of the slice do not fit the registers together: the two-pass pair on the kept statistics
                ws = _gn_workspace(x, N, ctx.groups)
                sums = torch.empty(N * ctx.groups * 2, dtype=torch.float32, device=x.device)
                _check(L.gd_nn_groupnorm_silu_backward(stream, x.data_ptr(), dy.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                       mr.data_ptr(), dx.data_ptr(), N, H * W, Cc, ctx.groups,
                                                       int(ctx.silu), ws.data_ptr(), sums.data_ptr(), None),
                       "gd_nn_groupnorm_silu_backward")
        return dx, None, None, None, None, None


def group_norm_silu(x, weight, bias, groups: int, eps: float, silu: bool = True):
    """``silu(group_norm(x))`` (or just group_norm).  HIP kernel for bf16 NHWC tensors on the GPU."""
    if x.is_cuda:
        if weight.requires_grad or bias.requires_grad:
            raise RuntimeError("group_norm_silu HIP kernel computes input gradients only (frozen weights)")
        if x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0:
            if not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
            small = lib().gd_nn_groupnorm_silu_fused_supported(x.shape[0], x.shape[2] * x.shape[3], x.shape[1], groups)
            if torch.is_grad_enabled() and x.requires_grad:
                if small and _GN_FUSED_TRAIN:
                    return _GroupNormSiLUSmall.apply(x, weight, bias, groups, eps, silu)
            elif small and _GN_FUSED_SMALL:
                return _gn_fused_small(x, weight, bias, groups, eps, silu)
            return _GroupNormSiLU.apply(x, weight, bias, groups, eps, silu)
        # fp32 GPU runs (parity checks of the bf16 path) use torch's ops
        _note_fallback("group_norm_silu", x, "needs a 4-D tensor with C % 8 == 0")
    y = F.group_norm(x, groups, weight, bias, eps)
    return F.silu(y) if silu else y


# ---------------------------------------------------------------------------------------------
# 3x3 convolution (implicit GEMM on MFMA), fused bias / per-image bias / residual
# ---------------------------------------------------------------------------------------------

def _bias_and_stride(bias):
    """[Cout] bias -> (bias, 0); per-image [N, Cout] bias -> (bias, elements between rows).  Column slices of a
    wider matrix (sd21.TembProjections) are passed as they are -- the kernels take the row stride."""
    if bias is None:
        return None, 0
    if bias.dim() == 1:
        return bias.contiguous(), 0
    if bias.stride(1) != 1 or bias.stride(0) % 8 or bias.storage_offset() % 8:
        bias = bias.contiguous()
    return bias, bias.stride(0)


# GD_NN_WINO=0: every stride-1 convolution on the direct kernels (A/B timing in tools/; never set in tests or the benchmark)
_WINO = os.environ.get("GD_NN_WINO", "1") != "0"

_ROUTE_SCALE = 1
_ROUTE_RANK = 0
_ROUTE_BATCH = None     # (groups, samples) of the call in flight, see route_batch
_ROUTE_OWNER = None     # weakref to the object that set the routing last (set_route_scale(owner=), release_route_scale)


def set_route_scale(k: int, rank: int = 0, owner=None) -> None:
    """Batch-invariant kernel selection (include/gd_nn.h gd_nn_conv_set_route_scale): every routing rule in this
    module and in the library that looks at the batch sees N * k images.  The sharded loop sets k = world size when
    asked for `batch_invariant` gradients, so a rank with 1/k of the views runs the kernels (and bf16 summation
    orders) the single-rank run of the whole camera batch runs; k = 1 tunes each launch for the batch it gets.
    `rank`: this rank's position among the k shares (dist.shard_views deals view v to rank v % k) -- the library GEMMs
    are given their rows at the positions the single-rank batch holds them (route_rows)."""
    global _ROUTE_SCALE, _ROUTE_RANK, _ROUTE_OWNER
    k, rank = int(k), int(rank)
    if not 0 <= rank < max(k, 1):
        raise ValueError(f"rank {rank} outside 0..{k - 1}")
    if lib().gd_nn_conv_set_route_scale(k) != 0:
        raise ValueError(f"route scale must be >= 1, got {k}")
    _ROUTE_SCALE, _ROUTE_RANK = k, rank
    _ROUTE_OWNER = weakref.ref(owner) if owner is not None else None


def release_route_scale(owner) -> None:
    """Back to k = 1 -- but only if `owner` is still the object that set the routing last (``set_route_scale(...,
    owner=)``): an old loop's close() / __del__ must not undo the setting of the loop that replaced it."""
    cur = _ROUTE_OWNER() if _ROUTE_OWNER is not None else None
    if cur is owner:
        set_route_scale(1, 0)


def route_scale() -> int:
    return _ROUTE_SCALE


class route_batch:
    """``with route_batch(groups, samples):`` -- the network call inside runs on `samples` batch entries made of
    `groups` concatenated copies of this rank's views (2 for the SDS UNet call: text | unconditional; 1 for the VAE).
    Only read under batch-invariant selection (set_route_scale(k > 1))."""

    def __init__(self, groups: int, samples: int):
        self.value = (int(groups), int(samples))

    def __enter__(self):
        global _ROUTE_BATCH
        self.prev, _ROUTE_BATCH = _ROUTE_BATCH, self.value
        return self

    def __exit__(self, *exc):
        global _ROUTE_BATCH
        _ROUTE_BATCH = self.prev
        return False


def route_rows(rows):
    """Under batch-invariant selection: the [M, K] row set of a library GEMM laid out as the single-rank run of the
    whole camera batch holds it -- k x M rows, this rank's samples at their global positions (view j of this rank is
    global view j * k + rank inside each of the `groups` copies), zeros elsewhere -- plus the function that takes this
    rank's rows back out of the product.  hipBLASLt picks tile and split-K from M, and (stream-K) the summation order of
    a row from the tile it falls in: both then match the single-rank run bit for bit."""
    k, r = _ROUTE_SCALE, _ROUTE_RANK
    M, K = rows.shape
    G, B = _ROUTE_BATCH if _ROUTE_BATCH is not None else (1, 0)
    if B <= 0 or B % G or M % B:
        G, B = 1, 1                      # unknown structure: this rank's M rows as block `rank` of k blocks, zeros elsewhere
    c, T = B // G, M // B
    padded = rows.new_zeros((G, c, k, T, K))
    padded[:, :, r] = rows.view(G, c, T, K)

    def take(out):
        return out.view(G, c, k, T, out.shape[-1])[:, :, r].reshape(M, out.shape[-1])
    return padded.view(k * M, K), take


def _conv_route(N, H, W, Cin, Cout, gn=False):
    """Which kernel a stride-1 3x3 convolution runs on: "wide" (128 channels x 16x32 pixels, csrc/nn_conv_wide.h), "wino"
    (Winograd F(2,3) along x, csrc/nn_conv_wino.h) or None = the direct patch-staged / implicit-GEMM kernels.  Distilled
    from tools/wino_route_bench.py and tools/gn_route_bench.py (every stride-1 shape of the SDS step, same box, against
    what the direct path picks): the wide tile wins where a tile has <= 128 output channels to work with (128 -> 128 @
    512^2 1.06x, 256 -> 128 1.08x, GroupNorm-fused 128 -> 128 1.08-1.11x) and on the 640-channel 32^2 level (1.13-1.21x);
    Winograd wins wherever the input is >= 320 channels deep (1.03-1.34x); few-input-channel layers that widen
    (128 -> 256, 256 -> 256, 256 -> 512: 0.90-1.02x) and every other GroupNorm-fused layer (0.87-0.91x) stay direct.
    Both need a grid of >= 128 tiles: small batches stay on the implicit-GEMM kernel's split-K."""
    if not _WINO or Cin % 32 or Cout % 8 or Cout < 64 or H < 16 or W < 16:
        return None
    tiles_n = (Cout + 127) // 128
    if Cout <= 128 and (not gn or Cin <= 128) or (not gn and Cout == 640 and W == 32 and Cin >= 320):
        # (the wide tile needs a fuller grid than the Winograd one: 256 -> 128 @ 256^2 at ONE image, 128 tiles, 0.72x)
        if N * _ROUTE_SCALE * ((H + 15) // 16) * ((W + 31) // 32) * tiles_n >= (256 if Cout <= 128 else 160) and \
                lib().gd_nn_conv3x3_wide_supported(N, H, W, Cin, Cout):
            return "wide"
    if gn or (Cin < 320 and Cout > 128):
        return None
    if N * _ROUTE_SCALE * ((H + 15) // 16) * ((W + 15) // 16) * tiles_n >= 128 and lib().gd_nn_conv3x3_wino_supported(N, H, W, Cin, Cout):
        return "wino"
    return None


def _conv_launch(x, w_khwc, bias, residual, out_channels):
    N, Cin, H, W = x.shape
    route = _conv_route(N, H, W, Cin, out_channels)
    if route == "wino":
        return _wino_launch(x, w_khwc, bias, residual, out_channels)
    if route == "wide":
        return _wide_launch(x, w_khwc, bias, residual, out_channels)
    L = lib()
    y = torch.empty((N, out_channels, H, W), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    bias, stride = _bias_and_stride(bias)
    # small-M layers run split over the taps and need fp32 scratch (torch's allocator: hipGraph-capture safe)
    ws_bytes = L.gd_nn_conv3x3_ws_bytes(N, H, W, Cin, out_channels)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        ret = L.gd_nn_conv3x3_forward_ws(stream, x.data_ptr(), w_khwc.data_ptr(),
                                         None if bias is None else bias.data_ptr(), stride,
                                         None if residual is None else residual.data_ptr(), y.data_ptr(), N, H, W, Cin,
                                         out_channels, None if ws is None else ws.data_ptr(), ws_bytes)
    if ret < 0:
        raise RuntimeError(f"gd_nn_conv3x3_forward failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
    return y


def _patch_launch(x, w_khwc, bias, residual, out_channels):
    """Plain 3x3/s1/p1 convolution on the patch-staged kernel (LDS-DMA patch, no GroupNorm)."""
    N, Cin, H, W = x.shape
    L = lib()
    y = torch.empty((N, out_channels, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    bias, stride = _bias_and_stride(bias)
    with torch.cuda.device(x.device):
        ret = L.gd_nn_conv3x3_gn_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), None, None, None,
                                         0, 0, w_khwc.data_ptr(), None if bias is None else bias.data_ptr(), stride,
                                         None if residual is None else residual.data_ptr(), y.data_ptr(), N, H, W, Cin,
                                         out_channels)
    if ret < 0:
        raise RuntimeError(f"gd_nn_conv3x3_gn_forward failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
    return y


def _wino(weight):
    """Cached Winograd F(2,3) filter bank (the kernel's 32 KB step images, include/gd_nn.h) of a frozen conv weight
    (channels_last, i.e. [Cout][3][3][Cin] in memory) -- also of a flipped dgrad weight tensor."""
    u = getattr(weight, "_gd_wino", None)
    key = (weight.data_ptr(), weight._version)
    if u is None or u.device != weight.device or getattr(weight, "_gd_wino_key", None) != key:
        Cout, Cin = weight.shape[0], weight.shape[1]
        u = torch.empty(lib().gd_nn_conv3x3_wino_weights_bytes(Cout, Cin) // 2, dtype=torch.bfloat16, device=weight.device)
        with torch.cuda.device(weight.device):
            ret = lib().gd_nn_conv3x3_wino_weights(torch.cuda.current_stream(weight.device).cuda_stream,
                                                   weight.data_ptr(), u.data_ptr(), Cout, Cin)
        _check(ret, "gd_nn_conv3x3_wino_weights", "gd_nn_conv_last_error")
        weight._gd_wino, weight._gd_wino_key = u, key
    return u


def _wino_launch(x, w_khwc, bias, residual, out_channels, stat_part=None):
    """Plain 3x3/s1/p1 convolution on the Winograd F(2,3)-along-x kernel (csrc/nn_conv_wino.h)."""
    N, Cin, H, W = x.shape
    L = lib()
    y = torch.empty((N, out_channels, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    bias, stride = _bias_and_stride(bias)
    u = _wino(w_khwc)
    with torch.cuda.device(x.device):
        ret = L.gd_nn_conv3x3_wino_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), u.data_ptr(),
                                           None if bias is None else bias.data_ptr(), stride,
                                           None if residual is None else residual.data_ptr(), y.data_ptr(), N, H, W, Cin,
                                           out_channels, None if stat_part is None else stat_part.data_ptr())
    _check(ret, "gd_nn_conv3x3_wino_forward", "gd_nn_conv_last_error")
    return y


def _wide(weight):
    """Cached re-packing of a frozen conv weight for the wide-tile kernel (24 KB step images, include/gd_nn.h)."""
    u = getattr(weight, "_gd_wide", None)
    key = (weight.data_ptr(), weight._version)
    if u is None or u.device != weight.device or getattr(weight, "_gd_wide_key", None) != key:
        Cout, Cin = weight.shape[0], weight.shape[1]
        u = torch.empty(lib().gd_nn_conv3x3_wide_weights_bytes(Cout, Cin) // 2, dtype=torch.bfloat16, device=weight.device)
        with torch.cuda.device(weight.device):
            ret = lib().gd_nn_conv3x3_wide_weights(torch.cuda.current_stream(weight.device).cuda_stream,
                                                   weight.data_ptr(), u.data_ptr(), Cout, Cin)
        _check(ret, "gd_nn_conv3x3_wide_weights", "gd_nn_conv_last_error")
        weight._gd_wide, weight._gd_wide_key = u, key
    return u


def _wide_launch(x, w_khwc, bias, residual, out_channels, stat_part=None, gn=None):
    """3x3/s1/p1 convolution on the 128-channel x 16x32-pixel tile (csrc/nn_conv_wide.h); gn = (mean_rstd, gamma, beta,
    groups, silu) applies GroupNorm(+SiLU) in the loader."""
    N, Cin, H, W = x.shape
    L = lib()
    y = torch.empty((N, out_channels, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    bias, stride = _bias_and_stride(bias)
    u = _wide(w_khwc)
    with torch.cuda.device(x.device):
        st = torch.cuda.current_stream(x.device).cuda_stream
        tail = (u.data_ptr(), None if bias is None else bias.data_ptr(), stride, None if residual is None else residual.data_ptr(),
                y.data_ptr(), N, H, W, Cin, out_channels, None if stat_part is None else stat_part.data_ptr())
        if gn is None:
            ret = L.gd_nn_conv3x3_wide_forward(st, x.data_ptr(), *tail)
        else:
            mr, gw, gb, groups, silu = gn
            ret = L.gd_nn_conv3x3_wide_gn_forward(st, x.data_ptr(), mr.data_ptr(), gw.data_ptr(), gb.data_ptr(), groups,
                                                  int(silu), *tail)
    _check(ret, "gd_nn_conv3x3_wide_forward", "gd_nn_conv_last_error")
    return y


def _wino_gn_launch(x, mean_rstd, gn_weight, gn_bias, groups, silu, w_khwc, bias, residual, out_channels, stat_part=None):
    """conv3x3(act(GroupNorm(x))) on the Winograd kernel, GroupNorm(+SiLU) applied in its loader; ``mean_rstd`` from
    gd_nn_groupnorm_stats / finish_partials."""
    N, Cin, H, W = x.shape
    L = lib()
    y = torch.empty((N, out_channels, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    bias, stride = _bias_and_stride(bias)
    u = _wino(w_khwc)
    with torch.cuda.device(x.device):
        ret = L.gd_nn_conv3x3_wino_gn_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(),
                                              mean_rstd.data_ptr(), gn_weight.data_ptr(), gn_bias.data_ptr(), groups,
                                              int(silu), u.data_ptr(), None if bias is None else bias.data_ptr(), stride,
                                              None if residual is None else residual.data_ptr(), y.data_ptr(), N, H, W, Cin,
                                              out_channels, None if stat_part is None else stat_part.data_ptr())
    _check(ret, "gd_nn_conv3x3_wino_gn_forward", "gd_nn_conv_last_error")
    return y


def _flipped(weight):
    """Cached dgrad weights [Cin][3][3][Cout] for a frozen conv weight (stored channels_last)."""
    f = getattr(weight, "_gd_flipped", None)
    key = (weight.data_ptr(), weight._version)      # re-derive after load_state_dict / in-place edits of the weight
    if f is None or f.device != weight.device or getattr(weight, "_gd_flipped_key", None) != key:
        Cout, Cin = weight.shape[0], weight.shape[1]
        f = torch.empty((Cin, Cout, 3, 3), dtype=torch.bfloat16, device=weight.device,
                        memory_format=torch.channels_last)
        L = lib()
        with torch.cuda.device(weight.device):
            ret = L.gd_nn_conv3x3_flip_weights(torch.cuda.current_stream(weight.device).cuda_stream,
                                               weight.data_ptr(), f.data_ptr(), Cout, Cin)
        if ret < 0:
            raise RuntimeError("gd_nn_conv3x3_flip_weights failed")
        weight._gd_flipped, weight._gd_flipped_key = f, key
    return f


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        ctx.weight = weight
        ctx.has_res = residual is not None
        ctx.bias_meta = None if bias is None else (bias.dim(), bias.dtype)
        return _conv_launch(x, weight, bias, residual, weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        w = ctx.weight
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = None
        if ctx.needs_input_grad[0]:
            if w.shape[0] % 64 == 0:
                dx = _conv_launch(dy, _flipped(w), None, None, w.shape[1])
            else:
                # tiny-Cout layers (the VAE's 512 -> 8 conv_out): the K of the dgrad GEMM (= Cout) is zero-padded to
                # one 64-channel K-step instead of handing the layer to MIOpen
                Cout, Cp = w.shape[0], (w.shape[0] + 63) // 64 * 64
                f = _flipped(w)
                key = (f.data_ptr(), w._version)
                fp = getattr(w, "_gd_flipped_pad", None)
                if fp is None or getattr(w, "_gd_flipped_pad_key", None) != key:
                    fp = torch.zeros((w.shape[1], Cp, 3, 3), dtype=torch.bfloat16, device=w.device).contiguous(
                        memory_format=torch.channels_last)
                    fp[:, :Cout] = f
                    w._gd_flipped_pad, w._gd_flipped_pad_key = fp, key
                dyp = torch.zeros((dy.shape[0], Cp) + tuple(dy.shape[2:]), dtype=torch.bfloat16, device=dy.device).contiguous(
                    memory_format=torch.channels_last)
                dyp[:, :Cout] = dy
                dx = _conv_launch(dyp, fp, None, None, w.shape[1])
        db = None
        if ctx.needs_input_grad[2]:
            # a bias that trains (the LoRA UNet's camera / shading embedding reaches every ResnetBlock2D's conv1 as its per-image
            # bias): the sum of dy over the pixels (and the images for a [Cout] bias), fp32 accumulation, one rounding
            dim, dt = ctx.bias_meta
            db = dy.sum(dim=(2, 3) if dim == 2 else (0, 2, 3), dtype=torch.float32).to(dt)
        return dx, None, db, (dy if ctx.has_res else None)


# GD_NN_FIRST_DGRAD=0: the first convolution's input gradient on the padded implicit-GEMM kernel as before round 5 (same-box A/B)
_FIRST_DGRAD = os.environ.get("GD_NN_FIRST_DGRAD", "1") != "0"


class _ConvSmallCin(torch.autograd.Function):
    """First VAE convolution (Cin = 3): forward on gd_nn_conv3x3_first_forward (matrix-core kernel for the VAE's 128
    output channels, VALU kernel otherwise; an output-write stream), input gradient through the MFMA kernel with the
    flipped weights zero-padded to 4 output channels (the library dgrad for this shape costs 2.7 ms per step on
    MI355X; this path ~0.6 ms).  ``next_norm`` = (groups, eps): GroupNorm statistics of the result from the kernel's
    epilogue, returned as a second (non-differentiable) output, or None when the shape has no such path."""

    @staticmethod
    def forward(ctx, x, weight, bias, next_norm=None):
        ctx.weight = weight
        ctx.x_shape = x.shape
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        if Cout % 8 == 0 and 36 * Cin * Cout <= 65536 and (bias is None or bias.dtype == torch.bfloat16):
            xc = x.contiguous(memory_format=torch.channels_last)
            wc = weight.contiguous(memory_format=torch.channels_last)
            y = torch.empty((N, Cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
            L = lib()
            rows = 0
            if next_norm is not None and _EPILOGUE_STATS and Cout % (4 * next_norm[0]) == 0:
                rows = L.gd_nn_conv3x3_first_stat_rows(N, H, W, Cin, Cout)
            mr_next = None
            with torch.cuda.device(x.device):
                stream = torch.cuda.current_stream(x.device).cuda_stream
                bias_p = None if bias is None else bias.data_ptr()
                if rows:
                    part = torch.empty(N * (Cout // 4) * rows * 2, dtype=torch.float32, device=x.device)
                    ret = L.gd_nn_conv3x3_first_forward_stats(stream, xc.data_ptr(), wc.data_ptr(), bias_p, y.data_ptr(),
                                                              N, H, W, Cin, Cout, part.data_ptr())
                else:
                    ret = L.gd_nn_conv3x3_first_forward(stream, xc.data_ptr(), wc.data_ptr(), bias_p, y.data_ptr(), N, H,
                                                        W, Cin, Cout)
                if ret < 0:
                    raise RuntimeError(f"gd_nn_conv3x3_first_forward failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
                if rows:
                    mr_next = torch.empty(N * next_norm[0] * 2, dtype=torch.float32, device=x.device)
                    _check(L.gd_nn_groupnorm_finish_partials(stream, part.data_ptr(), N, rows, Cout, next_norm[0], H * W,
                                                             float(next_norm[1]), mr_next.data_ptr()),
                           "gd_nn_groupnorm_finish_partials")
            if mr_next is not None:
                ctx.mark_non_differentiable(mr_next)
            return y, mr_next
        _note_fallback("conv3x3_small_cin", x, "needs Cout % 8 == 0, 36 * Cin * Cout <= 65536 and a bf16 bias")
        with torch.no_grad():
            return F.conv2d(x, weight, bias, padding=1), None

    @staticmethod
    def backward(ctx, dy, _dmr=None):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None
        w = ctx.weight
        Cout, Cin = w.shape[0], w.shape[1]
        N, _, H, W = dy.shape
        if _FIRST_DGRAD and dy.dtype == torch.bfloat16 and lib().gd_nn_conv3x3_first_dgrad_supported(N, H, W, Cin, Cout):
            # round 5: dy read once -- a 128 -> 9 x 3 product per pixel on the matrix cores, the nine shifted partial results summed
            # through LDS (csrc/nn_conv_first_dgrad.h); the padded implicit-GEMM form below read every dy element nine times
            wp = getattr(w, "_gd_first_dgrad", None)
            key = (w.data_ptr(), w._version)
            L = lib()
            dy = dy.contiguous(memory_format=torch.channels_last)
            with torch.cuda.device(w.device):
                stream = torch.cuda.current_stream(w.device).cuda_stream
                if wp is None or wp.device != w.device or getattr(w, "_gd_first_dgrad_key", None) != key:
                    wp = torch.empty(32 * 128, dtype=torch.bfloat16, device=w.device)
                    wc = w.contiguous(memory_format=torch.channels_last)
                    _check(L.gd_nn_conv3x3_first_dgrad_weights(stream, wc.data_ptr(), wp.data_ptr(), Cout, Cin),
                           "gd_nn_conv3x3_first_dgrad_weights", "gd_nn_conv_last_error")
                    w._gd_first_dgrad, w._gd_first_dgrad_key = wp, key
                dx4 = torch.empty((N, 4, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
                _check(L.gd_nn_conv3x3_first_dgrad(stream, dy.data_ptr(), wp.data_ptr(), dx4.data_ptr(), N, H, W, Cin, Cout),
                       "gd_nn_conv3x3_first_dgrad", "gd_nn_conv_last_error")
            return dx4[:, :Cin], None, None, None
        f = getattr(w, "_gd_flipped4", None)
        key = (w.data_ptr(), w._version)
        if f is None or f.device != w.device or getattr(w, "_gd_flipped4_key", None) != key:
            f = torch.zeros((4, Cout, 3, 3), dtype=torch.bfloat16, device=w.device).contiguous(
                memory_format=torch.channels_last)
            wc = w.contiguous(memory_format=torch.channels_last)
            with torch.cuda.device(w.device):
                ret = lib().gd_nn_conv3x3_flip_weights(torch.cuda.current_stream(w.device).cuda_stream, wc.data_ptr(),
                                                       f.data_ptr(), Cout, Cin)
            if ret < 0:
                raise RuntimeError("gd_nn_conv3x3_flip_weights failed")
            w._gd_flipped4, w._gd_flipped4_key = f, key
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx4 = _conv_launch(dy, f, None, None, 4)
        return dx4[:, :Cin], None, None, None


def conv3x3_small_cin(x, weight, bias, next_norm=None):
    """3x3/s1/p1 convolution with Cin <= 4 (image -> features).  ``next_norm``: the GroupNorm module that consumes the
    result, if known -- its statistics then ride on the returned tensor (see ``resnet_block_frozen``)."""
    if (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.shape[1] <= 4
            and weight.shape[0] % 64 == 0 and not weight.requires_grad):
        nn_ = None if next_norm is None else (next_norm.num_groups, next_norm.eps)
        y, mr_next = _ConvSmallCin.apply(x, weight, bias, nn_)
        if mr_next is not None:
            y._gd_gn_stats = (mr_next, nn_[0], nn_[1], y._version)
        return y
    _note_fallback("conv3x3_small_cin", x, "needs frozen bf16 weights with Cin <= 4 and Cout % 64 == 0")
    return F.conv2d(x, weight, bias, padding=1)


def conv1x1(x, weight, bias):
    """1x1 convolution on an NHWC tensor = a plain GEMM over the channel axis (hipBLASLt), no copies:
    the [N,H,W,C] view of a channels_last tensor is contiguous."""
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        y = F.linear(x.permute(0, 2, 3, 1), weight.flatten(1), bias)
        return y.permute(0, 3, 1, 2)
    _note_fallback("conv1x1", x, "needs a channels_last (NHWC) tensor")
    return F.conv2d(x, weight, bias)


def conv3x3_supported(x, weight) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] % 64 == 0 and weight.shape[0] % 4 == 0
            and weight.is_contiguous(memory_format=torch.channels_last))


def conv3x3(x, weight, bias=None, residual=None):
    """3x3 / stride 1 / pad 1 convolution (+ bias [Cout] or per-image bias [N,Cout]) (+ residual).
    MFMA implicit-GEMM HIP kernel for bf16 NHWC GPU tensors; torch ops otherwise (CPU / fp32)."""
    if conv3x3_supported(x, weight):
        if weight.requires_grad:
            raise RuntimeError("conv3x3 HIP kernel computes input and bias gradients only (frozen weights)")
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        return _Conv3x3.apply(x, weight, bias, residual)
    _note_fallback("conv3x3", x, "needs bf16 channels_last weights with Cin % 64 == 0 and Cout % 4 == 0")
    if bias is not None and bias.dim() == 2:
        y = F.conv2d(x, weight, None, padding=1) + bias[:, :, None, None]
    else:
        y = F.conv2d(x, weight, bias, padding=1)
    return y if residual is None else y + residual


def conv_profile(enable=None, reset=False):
    """Toggle / reset / read the conv kernel's event timing: returns (total_ms, launches, total_flops)."""
    L = lib()
    if reset:
        L.gd_nn_conv_profile_reset()
    if enable is not None:
        L.gd_nn_conv_profile_enable(int(enable))
    ms, n, fl = C.c_double(0), C.c_int64(0), C.c_double(0)
    L.gd_nn_conv_profile_read(C.byref(ms), C.byref(n), C.byref(fl))
    return ms.value, n.value, fl.value


def conv_profile_bytes() -> float:
    """Algorithmic HBM bytes (inputs + weights + outputs, each once) of the launches ``conv_profile`` has timed."""
    b = C.c_double(0)
    lib().gd_nn_conv_profile_read_bytes(C.byref(b))
    return b.value


class _GNConv3x3(torch.autograd.Function):
    """``conv3x3(act(group_norm(x))) + bias (+ residual)`` as statistics pass + ONE convolution kernel that
    normalises / activates in its activation loader (gd_nn_conv3x3_gn_forward).  Backward (frozen weights):
    dgrad convolution, then the GroupNorm(+SiLU) input-gradient kernel on the saved x and statistics."""

    @staticmethod
    def forward(ctx, x, gn_weight, gn_bias, groups, eps, silu, weight, bias, residual):
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        L = lib()
        ws = _gn_workspace(x, N, groups)
        mr = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
        y = torch.empty((N, Cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        gw, gb = gn_weight.contiguous(), gn_bias.contiguous()
        bias, stride = _bias_and_stride(bias)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _check(L.gd_nn_groupnorm_stats(stream, x.data_ptr(), N, H * W, Cin, groups, float(eps), ws.data_ptr(),
                                           mr.data_ptr()), "gd_nn_groupnorm_stats")
            ret = L.gd_nn_conv3x3_gn_forward(stream, x.data_ptr(), mr.data_ptr(), gw.data_ptr(), gb.data_ptr(), groups,
                                             int(silu), weight.data_ptr(), None if bias is None else bias.data_ptr(),
                                             stride, None if residual is None else residual.data_ptr(), y.data_ptr(),
                                             N, H, W, Cin, Cout)
        if ret < 0:
            raise RuntimeError(f"gd_nn_conv3x3_gn_forward failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
        ctx.save_for_backward(x, gw, gb, mr)
        ctx.weight, ctx.groups, ctx.silu, ctx.has_res = weight, groups, silu, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gw, gb, mr = ctx.saved_tensors
        w = ctx.weight
        N, Cin, H, W = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = None
        if ctx.needs_input_grad[0]:
            dact = _conv_launch(dy, _flipped(w), None, None, Cin)          # gradient w.r.t. act(GN(x))
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            ws = _gn_workspace(x, N, ctx.groups)
            sums = torch.empty(N * ctx.groups * 2, dtype=torch.float32, device=x.device)
            L = lib()
            with torch.cuda.device(x.device):
                stream = torch.cuda.current_stream(x.device).cuda_stream
                _check(L.gd_nn_groupnorm_silu_backward(stream, x.data_ptr(), dact.data_ptr(), gw.data_ptr(),
                                                       gb.data_ptr(), mr.data_ptr(), dx.data_ptr(), N, H * W, Cin,
                                                       ctx.groups, int(ctx.silu), ws.data_ptr(), sums.data_ptr(),
                                                       None),
                       "gd_nn_groupnorm_silu_backward")
        return dx, None, None, None, None, None, None, None, (dy if ctx.has_res else None)


def gn_conv3x3_supported(x, norm_weight, weight) -> bool:
    return (conv3x3_supported(x, weight) and weight.shape[0] % 64 == 0 and x.shape[1] % norm_weight.numel() == 0
            and not norm_weight.requires_grad)


def gn_conv3x3(x, norm_weight, norm_bias, groups: int, eps: float, silu: bool, weight, bias=None, residual=None):
    """Fused ``conv2d(act(group_norm(x)), weight, padding=1) + bias (+ residual)`` for frozen bf16 weights on the
    GPU; ``bias`` may be per image ([N, Cout])."""
    if not gn_conv3x3_supported(x, norm_weight, weight):
        raise RuntimeError("gn_conv3x3: unsupported tensor (need bf16 GPU, Cin % 64 == 0, Cout % 64 == 0)")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    w = weight if weight.is_contiguous(memory_format=torch.channels_last) else \
        weight.contiguous(memory_format=torch.channels_last)
    if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
        residual = residual.contiguous(memory_format=torch.channels_last)
    return _GNConv3x3.apply(x, norm_weight, norm_bias, groups, eps, silu, w, bias, residual)


_GN_FUSED_OVERRIDE = {"0": False, "1": True}.get(os.environ.get("GD_NN_GN_FUSED", ""))


def gn_conv_prefers_fused(x, out_channels: int) -> bool:
    """The largest feature maps (the VAE encoder's 512^2 and 256^2 levels) run GroupNorm + SiLU inside the
    patch-staged convolution's activation loader (1.07-1.17x faster than GroupNorm kernel + convolution on MI355X,
    tools/gn_conv_bench.py); on 128^2 and smaller maps, or below ~1.5 waves of workgroups (one per image, 16x16
    patch, 128/256-channel slab), the GroupNorm kernel + the persistent plain convolution is faster (0.95x)."""
    if _GN_FUSED_OVERRIDE is not None:      # GD_NN_GN_FUSED=0/1: A/B timing of the routing rule (tools/, never set in tests)
        return _GN_FUSED_OVERRIDE
    bn = 256 if out_channels % 256 == 0 else 128
    wgs = x.shape[0] * _ROUTE_SCALE * -(-x.shape[2] // 16) * -(-x.shape[3] // 16) * -(-out_channels // bn)
    return x.shape[2] * x.shape[3] >= 256 * 256 and wgs >= 384


def _gn_bwd_launch(x, dy, gw, gb, mr, groups, silu, add=None):
    N, Cc, H, W = x.shape
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    ws = _gn_workspace(x, N, groups)
    sums = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().gd_nn_groupnorm_silu_backward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(),
                                                   dy.data_ptr(), gw.data_ptr(), gb.data_ptr(), mr.data_ptr(),
                                                   dx.data_ptr(), N, H * W, Cc, groups, int(silu), ws.data_ptr(),
                                                   sums.data_ptr(), None if add is None else add.data_ptr()),
               "gd_nn_groupnorm_silu_backward")
    return dx


def _gnconv_forward(x, gw, gb, groups, eps, w, bias, residual, mr=None, next_norm=None):
    """``conv3x3(silu(group_norm(x))) + bias (+ residual)`` without autograd; returns (y, mean_rstd, next_mean_rstd).
    ``mr``: statistics of x already known (skips the statistics pass).  ``next_norm`` = (groups, eps) of a GroupNorm
    that consumes y: its statistics come out of this convolution's epilogue (gd_nn_conv3x3_*_stats +
    gd_nn_groupnorm_finish_partials) when the shape runs on a patch-staged kernel, else ``next_mean_rstd`` is None
    and the consumer runs its own statistics pass."""
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    L = lib()
    fused = gn_conv_prefers_fused(x, Cout)
    ws = _gn_workspace(x, N, groups) if mr is None else None
    have_mr = mr is not None
    if mr is None:
        mr = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
    rows = 0
    if next_norm is not None and _EPILOGUE_STATS and Cout % next_norm[0] == 0 and (Cout // next_norm[0]) % 4 == 0:
        rows = L.gd_nn_conv3x3_stat_rows(N, H, W, Cout, int(fused))
    part = torch.empty(N * (Cout // 4) * rows * 2, dtype=torch.float32, device=x.device) if rows else None
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream(x.device).cuda_stream
        y = torch.empty((N, Cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        bias_c, stride = _bias_and_stride(bias)
        bias_p = None if bias_c is None else bias_c.data_ptr()
        res_p = None if residual is None else residual.data_ptr()
        if fused:
            if not have_mr:
                _check(L.gd_nn_groupnorm_stats(stream, x.data_ptr(), N, H * W, Cin, groups, float(eps), ws.data_ptr(),
                                               mr.data_ptr()), "gd_nn_groupnorm_stats")
            if _conv_route(N, H, W, Cin, Cout, gn=True) == "wide":
                ret = L.gd_nn_conv3x3_wide_gn_forward(stream, x.data_ptr(), mr.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                                      groups, 1, _wide(w).data_ptr(), bias_p, stride, res_p, y.data_ptr(),
                                                      N, H, W, Cin, Cout, None if part is None else part.data_ptr())
            else:
                ret = L.gd_nn_conv3x3_gn_forward_stats(stream, x.data_ptr(), mr.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                                   groups, 1, w.data_ptr(), bias_p, stride, res_p, y.data_ptr(), N, H, W,
                                                   Cin, Cout, None if part is None else part.data_ptr())
            if ret < 0:
                raise RuntimeError(f"gd_nn_conv3x3_gn_forward failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
        else:
            act = torch.empty_like(x, memory_format=torch.channels_last)
            _check(L.gd_nn_groupnorm_silu_forward(stream, x.data_ptr(), act.data_ptr(), gw.data_ptr(), gb.data_ptr(), N,
                                                  H * W, Cin, groups, float(eps), 1, None if have_mr else ws.data_ptr(),
                                                  mr.data_ptr()), "gd_nn_groupnorm_silu_forward")
            if part is None:
                return _conv_launch(act, w, bias, residual, Cout), mr, None
            route = _conv_route(N, H, W, Cin, Cout)
            if route == "wino":
                ret = L.gd_nn_conv3x3_wino_forward(stream, act.data_ptr(), _wino(w).data_ptr(), bias_p, stride, res_p,
                                                   y.data_ptr(), N, H, W, Cin, Cout, part.data_ptr())
            elif route == "wide":
                ret = L.gd_nn_conv3x3_wide_forward(stream, act.data_ptr(), _wide(w).data_ptr(), bias_p, stride, res_p,
                                                   y.data_ptr(), N, H, W, Cin, Cout, part.data_ptr())
            else:
                ret = L.gd_nn_conv3x3_forward_stats(stream, act.data_ptr(), w.data_ptr(), bias_p, stride, res_p, y.data_ptr(),
                                                    N, H, W, Cin, Cout, part.data_ptr())
            if ret < 0:
                raise RuntimeError(f"gd_nn_conv3x3_forward_stats failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
        mr_next = None
        if part is not None:
            g2, eps2 = next_norm
            mr_next = torch.empty(N * g2 * 2, dtype=torch.float32, device=x.device)
            _check(L.gd_nn_groupnorm_finish_partials(stream, part.data_ptr(), N, rows, Cout, g2, H * W, float(eps2),
                                                     mr_next.data_ptr()), "gd_nn_groupnorm_finish_partials")
    return y, mr, mr_next


# GD_NN_EPILOGUE_STATS=0: every GroupNorm runs its own statistics pass (A/B timing of the epilogue statistics in
# tools/; never set in tests or the benchmark)
_EPILOGUE_STATS = os.environ.get("GD_NN_EPILOGUE_STATS", "1") != "0"


class _ResnetBlockFrozen(torch.autograd.Function):
    """diffusers ``ResnetBlock2D`` without time embedding (the VAE encoder's) as ONE autograd node:
    ``conv2(silu(norm2(conv1(silu(norm1(x)))))) + shortcut(x)``.  As separate nodes, x receives two gradients (main
    path and skip path) that autograd sums with an extra elementwise pass over the largest tensors of the step
    (1.2 ms per SDS step at 8 views); here the skip gradient goes into the GroupNorm backward of norm1
    (``add`` of gd_nn_groupnorm_silu_backward).  Saves x and conv1's output, as the separate nodes did."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, c1w, c1b, n2w, n2b, c2w, c2b, scw, scb, groups, eps, mr_in, next_norm):
        # GroupNorm statistics travel with the tensors: norm2's come out of conv1's epilogue, the next block's norm1's
        # (next_norm) out of conv2's, and this block's norm1 takes mr_in from its producer -- each saves a full read
        # of the activation (gd_nn.h, gd_nn_conv3x3_gn_forward_stats)
        h, mr1, mr2 = _gnconv_forward(x, n1w, n1b, groups, eps, c1w, c1b, None, mr=mr_in, next_norm=(groups, eps))
        skip = x
        if scw is not None:
            skip = F.linear(x.permute(0, 2, 3, 1), scw.flatten(1), scb).permute(0, 3, 1, 2)
        y, mr2, mr_next = _gnconv_forward(h, n2w, n2b, groups, eps, c2w, c2b, skip, mr=mr2, next_norm=next_norm)
        ctx.save_for_backward(x, h, mr1, mr2, n1w, n1b, n2w, n2b)
        ctx.c1w, ctx.c2w, ctx.scw, ctx.groups = c1w, c2w, scw, groups
        if mr_next is None:
            return y, None
        ctx.mark_non_differentiable(mr_next)
        return y, mr_next

    @staticmethod
    def backward(ctx, dy, _dmr=None):
        x, h, mr1, mr2, n1w, n1b, n2w, n2b = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dact2 = _conv_launch(dy, _flipped(ctx.c2w), None, None, ctx.c2w.shape[1])
        dh = _gn_bwd_launch(h, dact2, n2w, n2b, mr2, ctx.groups, True)
        dact1 = _conv_launch(dh, _flipped(ctx.c1w), None, None, ctx.c1w.shape[1])
        g_skip = dy
        if ctx.scw is not None:   # 1x1 shortcut: its input gradient is a plain GEMM on the NHWC view
            g_skip = torch.matmul(dy.permute(0, 2, 3, 1), ctx.scw.flatten(1)).permute(0, 3, 1, 2)
        dx = _gn_bwd_launch(x, dact1, n1w, n1b, mr1, ctx.groups, True, add=g_skip)
        return (dx,) + (None,) * 14


def resnet_block_frozen_supported(x, block) -> bool:
    """``block``: an sd21.ResnetBlock2D without time embedding, all parameters frozen, bf16 on the GPU, and x needs
    a gradient (without autograd the per-layer path has nothing to accumulate)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and torch.is_grad_enabled() and x.requires_grad):
        return False
    if block.time_emb_proj is not None or any(p.requires_grad for p in block.parameters()):
        return False
    c1, c2 = block.conv1.weight, block.conv2.weight
    return (conv3x3_supported(x, c1) and c1.shape[0] % 64 == 0 and c2.shape[0] % 64 == 0 and c2.shape[1] % 64 == 0
            and c2.is_contiguous(memory_format=torch.channels_last) and c1.shape[1] % block.norm1.num_groups == 0
            and c1.shape[1] % 8 == 0 and block.norm1.num_groups == block.norm2.num_groups
            and block.norm1.eps == block.norm2.eps)


def resnet_block_frozen(x, block, next_norm=None):
    """``next_norm``: the GroupNorm module that will consume the result (the next block's norm1), if the caller knows
    it: its statistics are then produced by this block's last convolution and ride on the returned tensor
    (``_gd_gn_stats``), where the next ``resnet_block_frozen`` call finds them."""
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    sc = block.conv_shortcut
    mr_in = None
    tag = getattr(x, "_gd_gn_stats", None)
    if tag is not None and tag[1:] == (block.norm1.num_groups, block.norm1.eps, x._version):
        mr_in = tag[0]
    nn_ = None if next_norm is None else (next_norm.num_groups, next_norm.eps)
    y, mr_next = _ResnetBlockFrozen.apply(x, block.norm1.weight, block.norm1.bias, block.conv1.weight, block.conv1.bias,
                                          block.norm2.weight, block.norm2.bias, block.conv2.weight, block.conv2.bias,
                                          None if sc is None else sc.weight, None if sc is None else sc.bias,
                                          block.norm1.num_groups, block.norm1.eps, mr_in, nn_)
    if mr_next is not None:
        y._gd_gn_stats = (mr_next, nn_[0], nn_[1], y._version)
    return y


def _up2_weights(weight):
    """Pre-summed filters of the upsample-fused convolution (include/gd_nn.h, gd_nn_conv3x3_up2_forward): two
    [Cout, Cin, 3, 3] channels_last tensors (even / odd output rows), summed in fp32, cached on the weight."""
    key = (weight.data_ptr(), weight._version)
    cached = getattr(weight, "_gd_up2", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    w = weight.detach().float()
    rows = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    out = []
    for py in (0, 1):
        slots = torch.zeros((weight.shape[0], weight.shape[1], 9), dtype=torch.float32, device=weight.device)
        for px in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    acc = 0
                    for ky in rows[(py, ty)]:
                        for kx in rows[(px, tx)]:
                            acc = acc + w[:, :, ky, kx]
                    slots[:, :, px * 4 + ty * 2 + tx] = acc
        out.append(slots.view(weight.shape[0], weight.shape[1], 3, 3).to(torch.bfloat16)
                   .contiguous(memory_format=torch.channels_last))
    weight._gd_up2 = (key, out[0], out[1])
    return out[0], out[1]


def upsample2x_conv3x3_supported(x, weight) -> bool:
    if not (conv3x3_supported(x, weight) and not weight.requires_grad and not torch.is_grad_enabled()):
        return False
    # each parity class is its own launch of N*H*W pixels x Cout channels in 128x128 tiles; below about half a
    # wave of tiles (one or two views per GPU) the single split-K launch on the upsampled tensor fills the chip better
    tiles = -(-(x.shape[0] * _ROUTE_SCALE * x.shape[2] * x.shape[3]) // 128) * -(-weight.shape[0] // 128)
    return tiles >= 128


def upsample2x_conv3x3(x, weight, bias=None):
    """``conv2d(interpolate(x, scale_factor=2, mode="nearest"), weight, bias, padding=1)`` for a frozen bf16 conv
    without autograd, as four 2x2-tap convolutions of x (2.25x fewer FLOPs, no upsampled tensor)."""
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    N, Cin, H, W = x.shape
    Cout = weight.shape[0]
    we, wo = _up2_weights(weight)
    y = torch.empty((N, Cout, 2 * H, 2 * W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_conv3x3_up2_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), we.data_ptr(),
                                          wo.data_ptr(), None if bias is None else bias.contiguous().data_ptr(),
                                          y.data_ptr(), N, H, W, Cin, Cout)
    if ret < 0:
        raise RuntimeError(f"gd_nn_conv3x3_up2_forward failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
    return y


class _Conv3x3S2(torch.autograd.Function):
    """3x3 / stride 2 / pad (pad_lo, 1) convolution, frozen weights: forward and input gradient on the MFMA kernel
    (the gradient as four parity-class launches)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad_lo):
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        Ho, Wo = (H + pad_lo - 2) // 2 + 1, (W + pad_lo - 2) // 2 + 1
        y = torch.empty((N, Cout, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        L = lib()
        ws_bytes = L.gd_nn_conv3x3_s2_ws_bytes(N, H, W, Cin, Cout, pad_lo, 0)     # small maps: split-K scratch
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
        with torch.cuda.device(x.device):
            ret = L.gd_nn_conv3x3_s2_forward_ws(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(),
                                                weight.data_ptr(), None if bias is None else bias.data_ptr(),
                                                y.data_ptr(), N, H, W, Cin, Cout, pad_lo,
                                                None if ws is None else ws.data_ptr(), ws_bytes)
        if ret < 0:
            raise RuntimeError(f"gd_nn_conv3x3_s2_forward failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
        ctx.weight, ctx.pad_lo, ctx.in_shape = weight, pad_lo, (N, Cin, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cin, H, W = ctx.in_shape
        Cout = ctx.weight.shape[0]
        dy = dy.contiguous(memory_format=torch.channels_last)
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = torch.empty((N, Cin, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
        wf = _flipped(ctx.weight)
        L = lib()
        ws_bytes = L.gd_nn_conv3x3_s2_ws_bytes(N, H, W, Cin, Cout, ctx.pad_lo, 1)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device) if ws_bytes else None
        with torch.cuda.device(dy.device):
            ret = L.gd_nn_conv3x3_s2_dgrad_ws(torch.cuda.current_stream(dy.device).cuda_stream, dy.data_ptr(),
                                              wf.data_ptr(), dx.data_ptr(), N, H, W, Cin, Cout, ctx.pad_lo,
                                              None if ws is None else ws.data_ptr(), ws_bytes)
        if ret < 0:
            raise RuntimeError(f"gd_nn_conv3x3_s2_dgrad failed ({ret}): {L.gd_nn_conv_last_error().decode()}")
        return dx, None, None, None


def conv3x3_s2_supported(x, weight) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4
            and weight.shape[2:] == (3, 3) and x.shape[1] % 64 == 0 and weight.shape[0] % 64 == 0
            and x.shape[2] >= 2 and x.shape[3] >= 2 and not weight.requires_grad)


def conv3x3_s2(x, weight, bias, pad_lo: int):
    """``conv2d(pad(x, (pad_lo, 1, pad_lo, 1)), weight, bias, stride=2)`` for frozen bf16 weights on the GPU."""
    if not conv3x3_s2_supported(x, weight):
        raise RuntimeError("conv3x3_s2: unsupported tensor (need bf16 GPU NHWC, Cin % 64 == 0, Cout % 64 == 0)")
    if bias is not None and bias.requires_grad:
        raise RuntimeError("conv3x3_s2 computes input gradients only (frozen weights)")
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    w = weight if weight.is_contiguous(memory_format=torch.channels_last) else \
        weight.contiguous(memory_format=torch.channels_last)
    return _Conv3x3S2.apply(x, w, bias, pad_lo)


# ---------------------------------------------------------------------------------------------
# transformer-block row passes (inference): GEGLU, residual add + LayerNorm
# ---------------------------------------------------------------------------------------------

def _rowwise_ok(*ts):
    return (not torch.is_grad_enabled() or not any(t.requires_grad for t in ts if t is not None)) and \
        all(t is None or (t.is_cuda and t.dtype == torch.bfloat16) for t in ts)


def _geglu_fwd(x, inner):
    x = x.contiguous()
    y = torch.empty(x.shape[:-1] + (inner,), dtype=x.dtype, device=x.device)
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_geglu_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), y.data_ptr(),
                                    x.numel() // (2 * inner), inner)
    if ret < 0:
        raise RuntimeError(f"gd_nn_geglu_forward failed ({ret}): {L.gd_nn_elementwise_last_error().decode()}")
    return x, y


class _GegluTrain(torch.autograd.Function):
    """GEGLU of the LoRA UNet's training pass: the forward kernel + ONE backward kernel that writes the gradient of the whole
    projection output [dh | dgate] (eager: chunk, gelu, mul forward; gelu_backward, two muls and a cat backward)."""

    @staticmethod
    def forward(ctx, x):
        inner = x.shape[-1] // 2
        xc, y = _geglu_fwd(x, inner)
        ctx.save_for_backward(xc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        inner = x.shape[-1] // 2
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L = lib()
        with torch.cuda.device(x.device):
            ret = L.gd_nn_geglu_backward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), dy.data_ptr(),
                                         dx.data_ptr(), x.numel() // (2 * inner), inner)
        if ret < 0:
            raise RuntimeError(f"gd_nn_geglu_backward failed ({ret}): {L.gd_nn_elementwise_last_error().decode()}")
        return dx


def softmax_rows_(s):
    """In-place softmax over the last dimension of a contiguous bf16 GPU tensor (gd_nn_softmax_rows_forward)."""
    L = s.shape[-1]
    with torch.cuda.device(s.device):
        ret = lib().gd_nn_softmax_rows_forward(torch.cuda.current_stream(s.device).cuda_stream, s.data_ptr(), s.data_ptr(),
                                               s.numel() // L, L)
    if ret < 0:
        raise RuntimeError(f"gd_nn_softmax_rows_forward failed ({ret}): {lib().gd_nn_elementwise_last_error().decode()}")
    return s


def softmax_rows_backward_(p, dp):
    """``dp <- p * (dp - sum(p * dp, -1))`` in place (gd_nn_softmax_rows_backward): the gradient of the scores."""
    L = p.shape[-1]
    with torch.cuda.device(p.device):
        ret = lib().gd_nn_softmax_rows_backward(torch.cuda.current_stream(p.device).cuda_stream, p.data_ptr(), dp.data_ptr(),
                                                dp.data_ptr(), p.numel() // L, L)
    if ret < 0:
        raise RuntimeError(f"gd_nn_softmax_rows_backward failed ({ret}): {lib().gd_nn_elementwise_last_error().decode()}")
    return dp


class _SingleHeadAttention(torch.autograd.Function):
    """``softmax(q k^T) v`` for ONE head of many channels (the VAE mid block: 4096 tokens x 512 channels per image) on the
    packed projection ``qkv`` [B, N, 3C] (the softmax scale already folded into the q rows): library GEMMs on the strided
    q / k / v views, the score matrix materialised in bf16 (34 MB per image -- cheaper here than a fused kernel whose
    backward for head_dim 512 runs far below the GEMMs), own row softmax forward and backward in place, and the three
    input gradients written by the GEMMs straight into the slices of ONE [B, N, 3C] tensor (autograd's split backward
    concatenated them: 148 us per step)."""

    @staticmethod
    def forward(ctx, qkv):
        Cc = qkv.shape[-1] // 3
        q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
        p = softmax_rows_(torch.bmm(q, k.transpose(1, 2)))
        ctx.save_for_backward(qkv, p)
        return torch.bmm(p, v)

    @staticmethod
    def backward(ctx, do):
        qkv, p = ctx.saved_tensors
        Cc = qkv.shape[-1] // 3
        q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
        do = do.contiguous()
        ds = softmax_rows_backward_(p, torch.bmm(do, v.transpose(1, 2)))
        dqkv = torch.empty_like(qkv)
        torch.bmm(ds, k, out=dqkv[..., :Cc])
        torch.bmm(ds.transpose(1, 2), q, out=dqkv[..., Cc:2 * Cc])
        torch.bmm(p.transpose(1, 2), do, out=dqkv[..., 2 * Cc:])
        return dqkv


def single_head_attention_supported(qkv) -> bool:
    return (qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and qkv.is_contiguous() and qkv.shape[-1] % 24 == 0
            and qkv.shape[1] % 8 == 0 and qkv.shape[1] <= 8192)


def single_head_attention(qkv):
    """[B, N, 3C] packed (scaled q | k | v) -> softmax(q k^T) v as [B, N, C]; see _SingleHeadAttention."""
    return _SingleHeadAttention.apply(qkv)


def _conv1x1_c8_launch(x, weight, bias, transposed):
    """x: [B, 8, H, W] bf16 in channels_last memory (NHWC) -> the same shape and layout."""
    y = torch.empty_like(x, memory_format=torch.channels_last)
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_conv1x1_c8(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), weight.data_ptr(),
                                 None if bias is None else bias.data_ptr(), y.data_ptr(), x.numel() // 8, int(transposed))
    if ret < 0:
        raise RuntimeError(f"gd_nn_conv1x1_c8 failed ({ret}): {L.gd_nn_elementwise_last_error().decode()}")
    return y


class _Conv1x1C8(torch.autograd.Function):
    """``F.conv2d(x, weight, bias)`` for the VAE's frozen quant_conv (8 -> 8 channels, 1x1): one pixel = one 16-byte vector
    forward, and the same kernel on the transposed weights for the input gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(weight)
        return _conv1x1_c8_launch(x, weight, bias, False)

    @staticmethod
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        return _conv1x1_c8_launch(dy.contiguous(memory_format=torch.channels_last), weight, None, True), None, None


def conv1x1_c8_supported(x, weight, bias) -> bool:
    return (x.is_cuda and x.dim() == 4 and x.shape[1] == 8 and x.dtype == torch.bfloat16 and tuple(weight.shape) == (8, 8, 1, 1)
            and weight.dtype == torch.bfloat16 and weight.is_contiguous() and not weight.requires_grad
            and (bias is None or (bias.dtype == torch.bfloat16 and bias.is_contiguous() and not bias.requires_grad))
            and x.data_ptr() % 16 == 0 and x.numel() > 0)


def conv1x1_c8(x, weight, bias=None):
    """The VAE's ``quant_conv`` on the own kernel (gd_nn_conv1x1_c8), with the input gradient; frozen weights only."""
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return _Conv1x1C8.apply(x, weight, bias)


_ATTN_TRAIN = os.environ.get("GD_ATTN_TRAIN", "1") != "0"    # A/B toggle: own attention forward kernel in the training pass
_ATTN_BWD = os.environ.get("GD_ATTN_BWD", "1") != "0"        # A/B toggle: own attention backward kernels (else the library's flash backward)
# Fewer keys than this: the library's flash backward.  Round 5: 1 -- the key-owning kernel cuts the query range into chunks
# when a head has one or two key blocks (the 77 text tokens), so the cross-attention backward is own code too
# (GD_ATTN_BWD_MIN_KEYS=256 restores the round-4 routing for same-box A/B)
_ATTN_BWD_MIN_KEYS = int(os.environ.get("GD_ATTN_BWD_MIN_KEYS", "1"))
_ROW_TRAIN = os.environ.get("GD_ROW_TRAIN", "1") != "0"    # A/B toggle: own GEGLU / LayerNorm backward in the training pass


def geglu(x):
    """``h * gelu(gate)`` with ``h, gate = x.chunk(2, -1)`` (diffusers GEGLU).  One HIP pass for bf16 GPU tensors; with a
    gradient (LoRA training) the same kernel under an autograd node whose backward is one kernel too.  The PyTorch ops
    otherwise (CPU, fp32)."""
    inner = x.shape[-1] // 2
    if x.is_cuda and x.dtype == torch.bfloat16 and inner % 8 == 0 and x.shape[-1] == 2 * inner:
        if _rowwise_ok(x):
            return _geglu_fwd(x, inner)[1]
        if _ROW_TRAIN:
            return _GegluTrain.apply(x)
    h, gate = x.chunk(2, dim=-1)
    return h * F.gelu(gate)


def _ln_backward(s, dy, weight, ds, eps):
    Cc = s.shape[-1]
    dx = torch.empty_like(s)
    L = lib()
    with torch.cuda.device(s.device):
        ret = L.gd_nn_layernorm_backward(torch.cuda.current_stream(s.device).cuda_stream, s.data_ptr(), dy.data_ptr(),
                                         weight.data_ptr(), None if ds is None else ds.data_ptr(), dx.data_ptr(),
                                         s.numel() // Cc, Cc, float(eps))
    if ret < 0:
        raise RuntimeError(f"gd_nn_layernorm_backward failed ({ret}): {L.gd_nn_elementwise_last_error().decode()}")
    return dx


def _add_ln_fwd(x, residual, weight, bias, eps, want_sum):
    Cc = x.shape[-1]
    x = x.contiguous()
    if residual is not None:
        residual = residual.contiguous()
    s = torch.empty_like(x) if (residual is not None and want_sum) else None
    y = torch.empty_like(x)
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_add_layernorm_forward(
            torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(),
            None if residual is None else residual.data_ptr(), weight.data_ptr(), bias.data_ptr(),
            float(eps), None if s is None else s.data_ptr(), y.data_ptr(), x.numel() // Cc, Cc)
    if ret < 0:
        raise RuntimeError(f"gd_nn_add_layernorm_forward failed ({ret}): {L.gd_nn_elementwise_last_error().decode()}")
    return (x if residual is None else s), y


class _AddLayerNormTrain(torch.autograd.Function):
    """``s = x + residual; y = LayerNorm(s)`` of the training pass (frozen norm parameters): the forward kernel, and ONE backward
    kernel for d s = (gradient reaching s through the residual stream) + LayerNorm-backward(dy), mean / rstd recomputed."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps):
        s, y = _add_ln_fwd(x, residual, weight, bias, eps, True)
        ctx.save_for_backward(s, weight)
        ctx.eps = eps
        ctx.has_res = residual is not None
        ctx.set_materialize_grads(False)
        if residual is None:
            return y
        return s, y

    @staticmethod
    def backward(ctx, *grads):
        s, weight = ctx.saved_tensors
        gs, gy = (None, grads[0]) if not ctx.has_res else grads
        if gy is None:
            d = gs
        else:
            d = _ln_backward(s, gy.contiguous(), weight, None if gs is None else gs.contiguous(), ctx.eps)
        return d, (d if ctx.has_res else None), None, None, None


def add_layer_norm(x, residual, norm, want_sum: bool = True):
    """``s = x + residual; return s, norm(s)`` (``residual`` None -> ``s = x``).  One HIP pass for bf16 GPU tensors; with a
    gradient on the activations and frozen norm parameters (LoRA training) the same kernel under an autograd node with a
    one-kernel backward; the PyTorch ops otherwise."""
    Cc = x.shape[-1]
    ok = Cc % 8 == 0 and Cc <= 2048 and norm.weight.dtype == torch.bfloat16 and x.is_cuda and x.dtype == torch.bfloat16 and \
        (residual is None or (residual.is_cuda and residual.dtype == torch.bfloat16))
    if ok and _rowwise_ok(x, residual, norm.weight, norm.bias):
        return _add_ln_fwd(x, residual, norm.weight, norm.bias, norm.eps, want_sum)
    if ok and _ROW_TRAIN and torch.is_grad_enabled() and not norm.weight.requires_grad and not norm.bias.requires_grad:
        if residual is None:
            return x, _AddLayerNormTrain.apply(x, None, norm.weight, norm.bias, float(norm.eps))
        return _AddLayerNormTrain.apply(x, residual, norm.weight, norm.bias, float(norm.eps))
    s = x if residual is None else x + residual
    return s, norm(s)


# ---------------------------------------------------------------------------------------------
# fused self-attention forward (head_dim 64, inference)
# ---------------------------------------------------------------------------------------------

def _attention_d64_layout_ok(q, k, v) -> bool:
    ok = lambda t: (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 4 and t.shape[-1] == 64 and t.stride(-1) == 1
                    and t.stride(2) == 64 and t.stride(1) % 8 == 0)   # noqa: E731
    return ok(q) and ok(k) and ok(v) and k.shape == v.shape and q.shape[0] == k.shape[0] and q.shape[2] == k.shape[2]


def attention_d64_supported(q, k, v) -> bool:
    """q, k, v: [B, S, H, 64] views (any batch / row stride, head and channel dims contiguous), no gradient needed."""
    return _attention_d64_layout_ok(q, k, v) and \
        not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad))


def attention_d64_train_supported(q, k, v) -> bool:
    """The same layouts WITH a gradient (training pass of the LoRA UNet): ``attention_d64_train``."""
    return _ATTN_TRAIN and torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad) and \
        _attention_d64_layout_ok(q, k, v)


def attention_d64(q, k, v):
    """softmax(q k^T / 8) v for [B, S, H, 64] views; returns [B, S, H*64] (contiguous).  Any number of keys: the kernel
    works on 64-key tiles and masks the padding (cross-attention over 77 text tokens)."""
    B, S, H, _ = q.shape
    kv_len = k.shape[1]
    Skv = (kv_len + 63) // 64 * 64
    L = lib()
    o = torch.empty((B, S, H * 64), dtype=torch.bfloat16, device=q.device)
    ws = torch.empty(L.gd_nn_attention_ws_bytes(B, Skv, H), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        ret = L.gd_nn_attention_d64_forward(torch.cuda.current_stream(q.device).cuda_stream, q.data_ptr(), k.data_ptr(),
                                            v.data_ptr(), o.data_ptr(), ws.data_ptr(), B, S, Skv, H, q.stride(0), q.stride(1),
                                            k.stride(0), k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1),
                                            64 ** -0.5, kv_len)
    if ret < 0:
        raise RuntimeError(f"gd_nn_attention_d64_forward failed ({ret}): {L.gd_nn_attention_last_error().decode()}")
    return o


class _AttentionD64Train(torch.autograd.Function):
    """Attention of the LoRA UNet's training pass: the own forward kernel (which also returns the per-query log-sum-exp) and the
    library's flash-attention backward, which recomputes the probabilities from exactly that tensor (natural log of the
    scaled scores' sum-exp, [B, H, S] fp32 -- tools/sdpa_lse_probe.py).  q, k, v: [B, S, H, 64] views."""

    @staticmethod
    def forward(ctx, q, k, v):
        B, S, H, _ = q.shape
        kv_len = k.shape[1]
        Skv = (kv_len + 63) // 64 * 64
        L = lib()
        o = torch.empty((B, S, H * 64), dtype=torch.bfloat16, device=q.device)
        lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
        ws = torch.empty(L.gd_nn_attention_ws_bytes(B, Skv, H), dtype=torch.uint8, device=q.device)
        with torch.cuda.device(q.device):
            ret = L.gd_nn_attention_d64_forward_lse(torch.cuda.current_stream(q.device).cuda_stream, q.data_ptr(), k.data_ptr(),
                                                    v.data_ptr(), o.data_ptr(), lse.data_ptr(), ws.data_ptr(), B, S, Skv, H,
                                                    q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
                                                    o.stride(0), o.stride(1), 64 ** -0.5, kv_len)
        if ret < 0:
            raise RuntimeError(f"gd_nn_attention_d64_forward_lse failed ({ret}): {L.gd_nn_attention_last_error().decode()}")
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, S, H, _ = q.shape
        kv_len = k.shape[1]
        # own backward kernels for self-attention; with a handful of keys (the 77 text tokens) the key-owning kernel has one
        # workgroup per head and the library's flash backward is faster (0.8x at S = 4096, tools/attn_bwd_bench.py)
        if _ATTN_BWD and S % 64 == 0 and kv_len >= _ATTN_BWD_MIN_KEYS:
            Skv = (kv_len + 63) // 64 * 64
            do = do.contiguous()
            L = lib()
            dq = torch.empty((B, S, H, 64), dtype=torch.bfloat16, device=q.device)
            dk = torch.empty((B, kv_len, H, 64), dtype=torch.bfloat16, device=q.device)
            dv = torch.empty((B, kv_len, H, 64), dtype=torch.bfloat16, device=q.device)
            ws = torch.empty(L.gd_nn_attention_bwd_ws_bytes(B, S, Skv, H), dtype=torch.uint8, device=q.device)
            with torch.cuda.device(q.device):
                ret = L.gd_nn_attention_d64_backward(
                    torch.cuda.current_stream(q.device).cuda_stream, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(),
                    do.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), B, S, Skv, H,
                    q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1),
                    do.stride(0), do.stride(1), dq.stride(0), dq.stride(1), dk.stride(0), dk.stride(1), dv.stride(0),
                    dv.stride(1), 64 ** -0.5, kv_len)
            if ret < 0:
                raise RuntimeError(f"gd_nn_attention_d64_backward failed ({ret}): {L.gd_nn_attention_last_error().decode()}")
            return dq, dk, dv
        z = torch.zeros((), dtype=torch.long, device=q.device)       # philox seed / offset: unused without dropout
        dq, dk, dv = torch.ops.aten._scaled_dot_product_flash_attention_backward(
            do.reshape(B, S, H, 64).transpose(1, 2), q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
            o.view(B, S, H, 64).transpose(1, 2), lse, None, None, S, k.shape[1], 0.0, False, z, z, scale=64 ** -0.5)
        return dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2)


def attention_d64_train(q, k, v):
    return _AttentionD64Train.apply(q, k, v)


def attention_d64_vt(q, k, vt):
    """The same kernel with V already transposed: ``vt`` = [B, H*64, Skv] contiguous bf16 (= W_v @ x^T), Skv % 64 == 0.
    q, k: [B, S, H, 64] views as in ``attention_d64``."""
    B, S, H, _ = q.shape
    Skv = k.shape[1]
    assert vt.shape == (B, H * 64, Skv) and vt.is_contiguous() and vt.dtype == torch.bfloat16 and Skv % 64 == 0
    L = lib()
    o = torch.empty((B, S, H * 64), dtype=torch.bfloat16, device=q.device)
    with torch.cuda.device(q.device):
        ret = L.gd_nn_attention_d64_forward_vt(torch.cuda.current_stream(q.device).cuda_stream, q.data_ptr(), k.data_ptr(),
                                               vt.data_ptr(), o.data_ptr(), B, S, Skv, H, q.stride(0), q.stride(1),
                                               k.stride(0), k.stride(1), o.stride(0), o.stride(1), 64 ** -0.5)
    if ret < 0:
        raise RuntimeError(f"gd_nn_attention_d64_forward_vt failed ({ret}): {L.gd_nn_attention_last_error().decode()}")
    return o


def attention_d64_vt_strided(q, k, vt, kv_len: int):
    """Cross-attention with V^T handed in as a [B, H*64, Skv] VIEW (rows contiguous, any batch stride: a channel slice of
    the all-layers context projection), ``kv_len`` <= Skv live keys (the rest of every V^T row is zero).  q: [B, S, H, 64],
    k: [B, kv_len, H, 64] views."""
    B, S, H, _ = q.shape
    Skv = vt.shape[2]
    assert vt.shape[:2] == (B, H * 64) and vt.stride(2) == 1 and vt.stride(1) == Skv and vt.dtype == torch.bfloat16 and Skv % 64 == 0
    L = lib()
    o = torch.empty((B, S, H * 64), dtype=torch.bfloat16, device=q.device)
    with torch.cuda.device(q.device):
        ret = L.gd_nn_attention_d64_forward_vt_strided(torch.cuda.current_stream(q.device).cuda_stream, q.data_ptr(), k.data_ptr(),
                                                       vt.data_ptr(), o.data_ptr(), B, S, Skv, H, q.stride(0), q.stride(1),
                                                       k.stride(0), k.stride(1), vt.stride(0), o.stride(0), o.stride(1),
                                                       64 ** -0.5, int(kv_len))
    if ret < 0:
        raise RuntimeError(f"gd_nn_attention_d64_forward_vt_strided failed ({ret}): {L.gd_nn_attention_last_error().decode()}")
    return o


# ---------------------------------------------------------------------------------------------------------------------
# image prologue of the guidance and the depth-sparsity head (csrc/nn_prologue.hip)
# ---------------------------------------------------------------------------------------------------------------------
class _VaePrologue(torch.autograd.Function):
    """``bf16_nhwc(2 * F.interpolate(x, (OH, OW), mode="bilinear", align_corners=False) - 1)`` as one launch forward and
    one backward (stable_diffusion_guidance.py:394-396,164 + the casts the VAE's first convolution needs)."""

    @staticmethod
    def forward(ctx, x, OH, OW):
        N, Cc, H, W = x.shape
        y = torch.empty((N, 3, OH, OW), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        L = lib()
        with torch.cuda.device(x.device):
            _check(L.gd_nn_vae_prologue_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), y.data_ptr(),
                                                N, H, W, OH, OW), "gd_nn_vae_prologue_forward", "gd_nn_prologue_last_error")
        ctx.shape = (N, H, W, OH, OW)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, OH, OW = ctx.shape
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        # the first convolution's input gradient comes on 4 zero-padded channels (a [:, :3] slice of an NHWC tensor):
        # read it in place, pixel stride CG
        base = dy._base if dy._base is not None and dy.storage_offset() == 0 else None
        if (base is not None and base.dtype == torch.bfloat16 and base.dim() == 4 and dy.shape[1] == 3
                and base.shape[0] == N and base.shape[1] >= 3 and base.shape[2:] == dy.shape[2:]
                and base.is_contiguous(memory_format=torch.channels_last) and dy.stride() == base.stride()
                and dy.data_ptr() == base.data_ptr()):     # dy is exactly base[:, :3]
            src, CG = base, base.shape[1]
        else:
            src, CG = dy.contiguous(memory_format=torch.channels_last), 3
        dx = torch.empty((N, 3, H, W), dtype=torch.float32, device=dy.device)
        L = lib()
        with torch.cuda.device(dy.device):
            _check(L.gd_nn_vae_prologue_backward(torch.cuda.current_stream(dy.device).cuda_stream, src.data_ptr(),
                                                 dx.data_ptr(), N, H, W, OH, OW, CG), "gd_nn_vae_prologue_backward", "gd_nn_prologue_last_error")
        return dx, None, None


def vae_prologue_supported(x) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous()


def vae_prologue(x, OH: int = 512, OW: int = 512):
    """x: planar fp32 [N,3,H,W] in [0,1] -> bf16 channels_last [N,3,OH,OW] = 2 * bilinear(x) - 1."""
    return _VaePrologue.apply(x, OH, OW)


class _SparsityHead(torch.autograd.Function):
    """``mean(sqrt((depth / (dmax + 1e-5))^2 + 0.01))`` (GaussianDreamer.py:215,253) given the maximum as a tensor:
    one kernel forward (value + d/d dmax), one backward."""

    @staticmethod
    def forward(ctx, depth, dmax):
        d = depth.contiguous()
        m = dmax.detach().reshape(1).to(torch.float32)
        sums = torch.empty(2, dtype=torch.float64, device=d.device)
        n = d.numel()
        L = lib()
        with torch.cuda.device(d.device):
            _check(L.gd_nn_sparsity_forward(torch.cuda.current_stream(d.device).cuda_stream, d.data_ptr(), m.data_ptr(), n,
                                            sums.data_ptr()), "gd_nn_sparsity_forward", "gd_nn_prologue_last_error")
        ctx.save_for_backward(d, m, sums)
        return (sums[0] / n).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        d, m, sums = ctx.saved_tensors
        n = d.numel()
        g32 = g.reshape(1).to(torch.float32).contiguous()
        dd = torch.empty_like(d)
        L = lib()
        with torch.cuda.device(d.device):
            _check(L.gd_nn_sparsity_backward(torch.cuda.current_stream(d.device).cuda_stream, d.data_ptr(), m.data_ptr(),
                                             g32.data_ptr(), n, dd.data_ptr()), "gd_nn_sparsity_backward", "gd_nn_prologue_last_error")
        dmax_grad = (-(sums[1] / n).to(torch.float32) / (m[0] + 1e-5)) * g32[0]
        return dd, dmax_grad.reshape(())


# ---------------------------------------------------------------------------------------------------------------------
# rank-4 LoRA branch (csrc/nn_lora.hip): y = base + scale * up(down(x)), forward and backward, fp32 accumulation
# ---------------------------------------------------------------------------------------------------------------------
def _lora_check(ret, what):
    if ret < 0:
        raise RuntimeError(f"{what} failed ({ret}): {lib().gd_nn_lora_last_error().decode()}")


def lora_branch_supported(x, base, down_w, up_w) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and base.dtype == torch.bfloat16 and down_w.dtype == torch.float32
            and up_w.dtype == torch.float32 and down_w.shape[0] == 4 and up_w.shape[1] == 4 and x.shape[-1] % 8 == 0
            and base.shape[-1] % 8 == 0 and down_w.is_contiguous() and up_w.is_contiguous())


def _lora_rowdot(a2d, w, scale, w_is_k_by_4):
    M, K = a2d.shape
    h = torch.empty((M, 4), dtype=torch.float32, device=a2d.device)
    with torch.cuda.device(a2d.device):
        _lora_check(lib().gd_nn_lora_rowdot(torch.cuda.current_stream(a2d.device).cuda_stream, a2d.data_ptr(), w.data_ptr(),
                                            h.data_ptr(), M, K, float(scale), int(w_is_k_by_4)), "gd_nn_lora_rowdot")
    return h


def _lora_rank4_add(h, w, base2d, N, w_is_n_by_4):
    M = h.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=h.device)
    with torch.cuda.device(h.device):
        _lora_check(lib().gd_nn_lora_rank4_add(torch.cuda.current_stream(h.device).cuda_stream, h.data_ptr(), w.data_ptr(),
                                               None if base2d is None else base2d.data_ptr(), y.data_ptr(), M, N,
                                               int(w_is_n_by_4)), "gd_nn_lora_rank4_add")
    return y


def _lora_colreduce(a2d, v, scale, g_is_j_by_4):
    M, J = a2d.shape
    L = lib()
    scratch = torch.empty(L.gd_nn_lora_colreduce_scratch_floats(M, J), dtype=torch.float32, device=a2d.device)
    g = torch.empty((J, 4) if g_is_j_by_4 else (4, J), dtype=torch.float32, device=a2d.device)
    with torch.cuda.device(a2d.device):
        _lora_check(L.gd_nn_lora_colreduce(torch.cuda.current_stream(a2d.device).cuda_stream, a2d.data_ptr(), v.data_ptr(),
                                           scratch.data_ptr(), g.data_ptr(), M, J, float(scale), int(g_is_j_by_4)),
                    "gd_nn_lora_colreduce")
    return g


_LORA_ROW_FUSED = os.environ.get("GD_LORA_ROW_FUSED", "1") != "0"


def _lora_row_fused(a2d, w1, w2, base2d, N, scale, backward, want_h=True):
    """One launch: h = scale * a . w1 and y = base + h . w2 (gd_nn_lora_row_fused; bit-identical to rowdot + rank4_add)."""
    M, K = a2d.shape
    if not _LORA_ROW_FUSED:     # same-box A/B only (tools/): the two launches the fused pass replaces, bit-identical
        h = _lora_rowdot(a2d, w1, scale, int(backward))
        return _lora_rank4_add(h, w2, base2d, N, int(not backward)), (h if want_h else None)
    h = torch.empty((M, 4), dtype=torch.float32, device=a2d.device) if want_h else None
    y = torch.empty((M, N), dtype=torch.bfloat16, device=a2d.device)
    with torch.cuda.device(a2d.device):
        _lora_check(lib().gd_nn_lora_row_fused(torch.cuda.current_stream(a2d.device).cuda_stream, a2d.data_ptr(), w1.data_ptr(),
                                               w2.data_ptr(), None if base2d is None else base2d.data_ptr(),
                                               None if h is None else h.data_ptr(), y.data_ptr(), M, K, N, float(scale),
                                               int(backward)), "gd_nn_lora_row_fused")
    return y, h


# ---- the weight gradients of all adapters of a backward pass as ONE grouped launch per stage (csrc/nn_lora.hip, round 6) ----
_LORA_GROUPING = os.environ.get("GD_LORA_GROUP", "1") != "0"     # =0: one reduction pair per adapter (same-box A/B)
_LORA_GROUP = [None]          # the group the adapted projections created right now belong to (lora_grad_group)
_LORA_GROUPS_OPEN = []        # groups with recorded but unlaunched problems (weakrefs): lora_groups_pending()


class LoraGradGroup:
    """Collects the (dy, scale * h, x, dh) -> (d up, d down) problems of the adapted projections of ONE forward pass whose
    gradients land in FlatAdam's sinks (nobody reads them before the optimizer step) and launches them together once the LAST
    of them has been recorded -- every projection created inside ``lora_grad_group()`` counts, each records itself in its
    backward.  Captured inside a hipGraph like any other launch: the table travels by value in the kernel arguments, 32
    adapters per launch."""

    def __init__(self):
        self.expected = 0
        self.entries = []
        self.keep = []
        self.grid = [0, 0, 0]
        self.launches = 0

    def add(self, dy, hs, x2d, dh, scratch, sink_up, sink_down, M, N, K):
        L = lib()
        eb = L.gd_nn_lora_colreduce_group_entry_bytes()
        buf = (C.c_ubyte * eb)()
        g3 = (C.c_int * 3)()
        _lora_check(L.gd_nn_lora_colreduce_group_desc(buf, dy.data_ptr(), hs.data_ptr(), x2d.data_ptr(), dh.data_ptr(),
                                                      scratch.data_ptr(), sink_up.data_ptr(), sink_down.data_ptr(), M, N, K, 1, g3),
                    "gd_nn_lora_colreduce_group_desc")
        if not self.entries:
            _LORA_GROUPS_OPEN.append(weakref.ref(self))
        self.entries.append(bytes(buf))
        self.keep.append((dy, hs, x2d, dh, scratch))          # alive until the grouped launch has been queued
        self.grid = [max(a, int(b)) for a, b in zip(self.grid, g3)]
        if len(self.entries) == self.expected:
            self.flush(dy.device)

    def flush(self, device):
        if not self.entries:
            return
        n = len(self.entries)
        table = (C.c_ubyte * (n * len(self.entries[0]))).from_buffer_copy(b"".join(self.entries))
        L = lib()
        with torch.cuda.device(device):
            _lora_check(L.gd_nn_lora_colreduce_group_launch(torch.cuda.current_stream(device).cuda_stream, table, n,
                                                            self.grid[0], self.grid[1], self.grid[2]),
                        "gd_nn_lora_colreduce_group_launch")
        self.entries, self.keep, self.grid = [], [], [0, 0, 0]
        self.launches += 1


class lora_grad_group:
    """``with lora_grad_group():`` around a forward pass under autograd: the adapted projections created inside send their
    weight gradients through ONE grouped launch per stage at the end of the backward pass (LoraGradGroup)."""

    def __enter__(self):
        self.prev = _LORA_GROUP[0]
        _LORA_GROUP[0] = self.group = LoraGradGroup() if _LORA_GROUPING else None
        return self.group

    def __exit__(self, *exc):
        _LORA_GROUP[0] = self.prev
        return False


def lora_groups_pending() -> int:
    """Recorded but unlaunched weight-gradient problems (a backward pass that did not reach every adapted projection of its
    forward pass): flat_adam.FlatAdam.step refuses to step over them."""
    n = 0
    for r in list(_LORA_GROUPS_OPEN):
        g = r()
        if g is None or not g.entries:
            _LORA_GROUPS_OPEN.remove(r)
        else:
            n += len(g.entries)
    return n


def _grad_sinks(down_w, up_w):
    """(sink of down, sink of up): fp32 tensors of the weights' shapes that RECEIVE the weight gradients in place
    (``Parameter._gd_grad_sink``, set by flat_adam.FlatAdam), or (None, None)."""
    sd, su = getattr(down_w, "_gd_grad_sink", None), getattr(up_w, "_gd_grad_sink", None)
    if sd is None or su is None or sd.shape != down_w.shape or su.shape != up_w.shape or not sd.is_contiguous() or \
            not su.is_contiguous():
        return None, None
    return sd, su


def _join_group(ctx, down_w, up_w):
    """The node joins the open LoraGradGroup if its weight gradients go to sinks and both adapter halves train."""
    grp = _LORA_GROUP[0]
    ctx.group = None
    if grp is not None and ctx.sinks[0] is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]:
        ctx.group = grp
        grp.expected += 1


def _lora_backward(ctx, dy, dx_base, need_x, need_down, need_up):
    """The adapter's share of the backward pass: dx (= dx_base + dh . down if a frozen projection's gradient is handed in),
    d down, d up.  Training case (both adapter halves): 3 launches -- the fused row pass, the paired reduction, its finish."""
    x2d, hs, down_w, up_w = ctx.saved_tensors[:4]
    dx = d_down = d_up = None
    K = x2d.shape[1]
    dh = None
    if need_x:
        dx, dh = _lora_row_fused(dy, up_w, down_w, dx_base, K, ctx.scale, 1, want_h=need_down)
    elif need_down:
        dh = _lora_rowdot(dy, up_w, ctx.scale, 1)                      # scale * dy @ up
    if need_up and need_down:        # both weight gradients from one launch per stage
        M, N = dy.shape
        L = lib()
        scratch = torch.empty(L.gd_nn_lora_colreduce_pair_scratch_floats(M, N, K), dtype=torch.float32, device=dy.device)
        sink_down, sink_up = getattr(ctx, "sinks", (None, None))
        if sink_down is not None and sink_up is not None:
            # the adapters' .grad are slices of ONE flat gradient buffer (flat_adam.FlatAdam): added in place by the kernel,
            # nothing returned to autograd -- no 256 gradient tensors, AccumulateGrad calls and gathers per UNet backward
            grp = getattr(ctx, "group", None)
            if grp is not None:            # ... and nobody reads them before the optimizer step: one grouped launch at the end
                grp.add(dy, hs, x2d, dh, scratch, sink_up, sink_down, M, N, K)
                return dx, None, None
            with torch.cuda.device(dy.device):
                _lora_check(L.gd_nn_lora_colreduce_pair_into(torch.cuda.current_stream(dy.device).cuda_stream, dy.data_ptr(),
                                                             hs.data_ptr(), x2d.data_ptr(), dh.data_ptr(), scratch.data_ptr(),
                                                             sink_up.data_ptr(), sink_down.data_ptr(), M, N, K, 1),
                            "gd_nn_lora_colreduce_pair_into")
            return dx, None, None
        d_up = torch.empty((N, 4), dtype=torch.float32, device=dy.device)
        d_down = torch.empty((4, K), dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            _lora_check(L.gd_nn_lora_colreduce_pair(torch.cuda.current_stream(dy.device).cuda_stream, dy.data_ptr(), hs.data_ptr(),
                                                    x2d.data_ptr(), dh.data_ptr(), scratch.data_ptr(), d_up.data_ptr(),
                                                    d_down.data_ptr(), M, N, K), "gd_nn_lora_colreduce_pair")
    elif need_up:
        d_up = _lora_colreduce(dy, hs, 1.0, 1)                         # [N, 4]: sum_m dy[m, n] * scale * h[m, r]
    elif need_down:
        d_down = _lora_colreduce(x2d, dh, 1.0, 0)                      # [4, K]
    return dx, d_down, d_up


class _LoraBranch(torch.autograd.Function):
    """y = base + scale * (x @ down^T) @ up^T with x [M, K] / base [M, N] bf16 and fp32 rank-4 adapters: ONE launch forward
    (four dot products per row, then four FMAs per output element on top of ``base``, by the same wave), three backward --
    every sum in fp32."""

    @staticmethod
    def forward(ctx, x2d, base2d, down_w, up_w, scale):
        y, hs = _lora_row_fused(x2d, down_w, up_w, base2d, base2d.shape[1], scale, 0)   # hs = scale * down(x)
        ctx.save_for_backward(x2d, hs, down_w, up_w)
        ctx.scale = float(scale)
        ctx.sinks = _grad_sinks(down_w, up_w)
        _join_group(ctx, down_w, up_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        need_x, need_base, need_down, need_up = ctx.needs_input_grad[:4]
        dx, d_down, d_up = _lora_backward(ctx, dy, None, need_x, need_down, need_up)
        return dx, (dy if need_base else None), d_down, d_up, None


class _LoraLinear(torch.autograd.Function):
    """The adapted projection as ONE autograd node: y = base_fn(x) + scale * up(down(x)) with the frozen projection inside
    (``base_fn`` runs under no_grad, so the inference routing applies -- the own K = 320 streaming GEMM on the 64x64-token
    blocks).  Backward: the frozen projection's dx = dy @ W from the library, handed to the fused row pass as its base, so
    the input gradient leaves complete -- autograd's add of the two branches' gradients (one elementwise pass per adapted
    projection, 128 per UNet backward) is gone."""

    @staticmethod
    def forward(ctx, x2d, weight, down_w, up_w, scale, base_fn):
        base2d = base_fn(x2d)
        y, hs = _lora_row_fused(x2d, down_w, up_w, base2d, base2d.shape[1], scale, 0)
        ctx.save_for_backward(x2d, hs, down_w, up_w, weight)
        ctx.scale = float(scale)
        ctx.sinks = _grad_sinks(down_w, up_w)
        _join_group(ctx, down_w, up_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        need_x, _, need_down, need_up = ctx.needs_input_grad[:4]
        dx_base = torch.mm(dy, ctx.saved_tensors[4]) if need_x else None
        dx, d_down, d_up = _lora_backward(ctx, dy, dx_base, need_x, need_down, need_up)
        return dx, None, d_down, d_up, None, None


def lora_linear(x, weight, down_w, up_w, scale, base_fn):
    """``base_fn(x) + scale * up(down(x))`` where ``base_fn`` is the FROZEN projection ``F.linear(., weight, bias)`` (any routing of
    it): one autograd node, complete input gradient (see _LoraLinear)."""
    assert not weight.requires_grad
    x2d = x.reshape(-1, x.shape[-1])
    if not x2d.is_contiguous():
        x2d = x2d.contiguous()
    y = _LoraLinear.apply(x2d, weight, down_w, up_w, scale, base_fn)
    return y.view(*x.shape[:-1], y.shape[-1])


def lora_branch(x, base, down_w, up_w, scale):
    """``base + scale * up(down(x))`` for [..., K] / [..., N] bf16 activations (the leading dimensions must agree)."""
    x2d = x.reshape(-1, x.shape[-1])
    b2d = base.reshape(-1, base.shape[-1])
    if not x2d.is_contiguous():
        x2d = x2d.contiguous()
    if not b2d.is_contiguous():
        b2d = b2d.contiguous()
    return _LoraBranch.apply(x2d, b2d, down_w, up_w, scale).view(base.shape)


def sparsity_loss(depth, dmax):
    """depth: fp32 CUDA tensor (any shape), dmax: 0-d tensor = its (global) maximum, attached to the graph."""
    if depth.is_cuda and depth.dtype == torch.float32:
        return _SparsityHead.apply(depth, dmax.reshape(()))
    return ((depth / (dmax + 1e-5)) ** 2 + 0.01).sqrt().mean()


# ---------------------------------------------------------------------------------------------------------------------
# fp8 (OCP e4m3) path of the no-grad UNet forward (csrc/nn_fp8.hip)
# ---------------------------------------------------------------------------------------------------------------------
FP8_MAX = 448.0


def _check8(ret, what):
    if ret < 0:
        raise RuntimeError(f"{what} failed ({ret}): {lib().gd_nn_fp8_last_error().decode()}")


def fp8_quantize(x, scale: float):
    """bf16 tensor -> e4m3 bytes (uint8 tensor of the same shape), value = scale * byte."""
    xc = x.contiguous()
    y = torch.empty(xc.shape, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _check8(lib().gd_nn_fp8_quantize(torch.cuda.current_stream(x.device).cuda_stream, xc.data_ptr(), y.data_ptr(),
                                         xc.numel(), 1.0 / scale), "gd_nn_fp8_quantize")
    return y


def fp8_pack_weights(w2d, scale: float):
    """bf16 [rows, K] -> e4m3 [rows, Kp] with Kp = K rounded up to 128 (zero padded)."""
    rows, K = w2d.shape
    Kp = (K + 127) // 128 * 128
    wc = w2d.contiguous()
    out = torch.empty((rows, Kp), dtype=torch.uint8, device=w2d.device)
    with torch.cuda.device(w2d.device):
        _check8(lib().gd_nn_fp8_pack_weights(torch.cuda.current_stream(w2d.device).cuda_stream, wc.data_ptr(),
                                             out.data_ptr(), rows, K, Kp, 1.0 / scale), "gd_nn_fp8_pack_weights")
    return out


def fp8_linear(x8, w8, bias, residual, K: int, dq: float):
    """x8: e4m3 [..., K], w8: e4m3 [Nout, Kp] -> bf16 [..., Nout] = dq * x8 . w8^T + bias + residual."""
    Nout, Kp = w8.shape
    M = x8.numel() // K
    y = torch.empty(x8.shape[:-1] + (Nout,), dtype=torch.bfloat16, device=x8.device)
    p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    with torch.cuda.device(x8.device):
        _check8(lib().gd_nn_fp8_linear_forward(torch.cuda.current_stream(x8.device).cuda_stream, x8.data_ptr(), w8.data_ptr(),
                                               p(bias), p(residual), y.data_ptr(), M, K, Kp, Nout, dq),
                "gd_nn_fp8_linear_forward")
    return y


def fp8_conv3x3(x8, w8, bias, residual, Cin: int, dq: float):
    """x8: e4m3 NHWC bytes as a channels_last uint8 tensor [N, Cin, H, W]; w8: e4m3 [Cout * 9, CinP];
    bias: bf16 [Cout] or per image [N, Cout]; residual: bf16 channels_last [N, Cout, H, W]."""
    N, _, H, W = x8.shape
    CinP = w8.shape[1]
    Cout = w8.shape[0] // 9
    y = torch.empty((N, Cout, H, W), dtype=torch.bfloat16, device=x8.device, memory_format=torch.channels_last)
    b, bs = _bias_and_stride(bias)
    with torch.cuda.device(x8.device):
        _check8(lib().gd_nn_fp8_conv3x3_forward(torch.cuda.current_stream(x8.device).cuda_stream, x8.data_ptr(),
                                                w8.data_ptr(), None if b is None else b.data_ptr(), bs,
                                                None if residual is None else residual.data_ptr(), y.data_ptr(), N, H, W,
                                                Cin, CinP, Cout, dq), "gd_nn_fp8_conv3x3_forward")
    return y


def group_norm_silu_fp8(x, weight, bias, groups: int, eps: float, silu: bool, scale: float):
    """``e4m3(act(GN(x)) / scale)`` of a bf16 channels_last tensor as a channels_last uint8 tensor (inference only)."""
    N, Cc, H, W = x.shape
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    y = torch.empty((N, Cc, H, W), dtype=torch.uint8, device=x.device, memory_format=torch.channels_last)
    ws = _gn_workspace(x, N, groups)
    mr = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
    w, b = weight.contiguous(), bias.contiguous()
    with torch.cuda.device(x.device):
        _check(lib().gd_nn_groupnorm_silu_forward_fp8(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(),
                                                      y.data_ptr(), w.data_ptr(), b.data_ptr(), N, H * W, Cc, groups,
                                                      float(eps), int(silu), ws.data_ptr(), mr.data_ptr(), 1.0 / scale),
               "gd_nn_groupnorm_silu_forward_fp8")
    return y


class Fp8State:
    """Static per-tensor e4m3 quantisation of the frozen UNet's 3x3 convolutions for the no-grad forward.

    ``mode == "calibrate"``: the forward runs in bf16 and records the largest |activation| every fp8 site sees;
    ``mode == "run"``: sites whose shape profits (tools/fp8_bench.py: everything but the 8x8 maps) run
    GroupNorm+SiLU -> e4m3 (one kernel) and the fp8 implicit-GEMM convolution with scale = margin * amax / 448 for
    the activations and |w|max / 448 for the (cached) weights."""

    def __init__(self, margin: float = 2.0, min_pixels: int = 2048):
        self.mode = "calibrate"
        self.margin, self.min_pixels = margin, min_pixels
        self.amax = {}       # id(conv) -> python float
        self.weights = {}    # id(conv) -> (w8, w_scale, weight version)
        self.sites_run = 0

    def wants(self, conv, x) -> bool:
        N, Cin, H, W = x.shape
        return (x.is_cuda and x.dtype == torch.bfloat16 and N * _ROUTE_SCALE * H * W >= self.min_pixels and Cin % 16 == 0
                and conv.out_channels % 4 == 0 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
                and conv.padding == (1, 1))

    def observe(self, conv, act):
        self.amax[id(conv)] = max(self.amax.get(id(conv), 0.0), float(act.detach().abs().max()))

    def _weights(self, conv):
        w = conv.weight
        ent = self.weights.get(id(conv))
        if ent is None or ent[2] != w._version:
            ws = max(float(w.detach().abs().max()), 1e-12) / FP8_MAX
            w2d = w.detach().to(torch.bfloat16).permute(0, 2, 3, 1).reshape(w.shape[0] * 9, w.shape[1])   # [Cout][3][3][Cin]
            ent = self.weights[id(conv)] = (fp8_pack_weights(w2d, ws), ws, w._version)
        return ent

    def gn_conv(self, norm, conv, x, bias, residual):
        sx = max(self.amax[id(conv)], 1e-6) * self.margin / FP8_MAX
        w8, ws, _ = self._weights(conv)
        x8 = group_norm_silu_fp8(x, norm.weight, norm.bias, norm.num_groups, norm.eps, True, sx)
        self.sites_run += 1
        return fp8_conv3x3(x8, w8, bias, residual, x.shape[1], sx * ws)


def linear_supported(x, weight) -> bool:
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.is_contiguous()
            and weight.is_contiguous() and weight.shape[1] % 64 == 0 and weight.shape[0] % 4 == 0
            and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)))


def linear_320_supported(x, weight, bias=None) -> bool:
    """bf16 GPU rows, K = 320, N = 320 / 640 / 2560, at least 4096 rows (the 64x64-token transformer blocks at any
    batch); x and weight 16-byte aligned (LDS-DMA / 16-byte fragment loads), bias bf16 and 8-byte aligned (uint2 loads)
    -- a view at an odd element offset falls back to ``F.linear`` instead of being read misaligned."""
    if bias is not None and (bias.dtype != torch.bfloat16 or not bias.is_contiguous() or bias.data_ptr() % 8):
        return False
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.is_contiguous()
            and weight.is_contiguous() and weight.dim() == 2 and x.shape[-1] == weight.shape[1]
            and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0
            and bool(lib().gd_nn_linear_320_supported(x.numel() // x.shape[-1], x.shape[-1], weight.shape[0])))


def linear_320(x, weight, bias=None):
    """``F.linear(x, weight, bias)`` for K = 320 (N = 320, 640, 2560) on the weights-in-registers streaming kernel,
    inference only."""
    M, N = x.numel() // 320, weight.shape[0]
    y = torch.empty(x.shape[:-1] + (N,), dtype=torch.bfloat16, device=x.device)
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_linear_k320_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), weight.data_ptr(),
                                          None if bias is None else bias.data_ptr(), y.data_ptr(), M, N)
    if ret < 0:
        raise RuntimeError(f"gd_nn_linear_k320_forward failed ({ret}): {L.gd_nn_linear_320_last_error().decode()}")
    return y


def linear_320_geglu(x, weight, bias=None):
    """diffusers ``GEGLU(320, 1280)``: ``h, g = F.linear(x, weight, bias).chunk(2, -1); h * gelu(g)`` as ONE kernel (the
    GEGLU arithmetic in the streaming GEMM's store phase); bit-identical to ``geglu(linear_320(x, weight, bias))``."""
    M, inner = x.numel() // 320, weight.shape[0] // 2
    y = torch.empty(x.shape[:-1] + (inner,), dtype=torch.bfloat16, device=x.device)
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_linear_k320_geglu_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(),
                                                weight.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                                M, inner)
    if ret < 0:
        raise RuntimeError(f"gd_nn_linear_k320_geglu_forward failed ({ret}): {L.gd_nn_linear_320_last_error().decode()}")
    return y


def _gemm_ok(x, weight, bias=None, residual=None) -> bool:
    K = weight.shape[1]
    ok = (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.is_contiguous()
          and weight.is_contiguous() and weight.dim() == 2 and x.shape[-1] == K and x.data_ptr() % 16 == 0
          and weight.data_ptr() % 16 == 0)
    for t, al in ((bias, 16), (residual, 8)):
        if t is not None and (t.dtype != torch.bfloat16 or not t.is_contiguous() or t.data_ptr() % al):
            return False
    return bool(ok)


def gemm_supported(x, weight, bias=None, residual=None, geglu: bool = False) -> bool:
    """bf16 contiguous GPU operands, K % 64 == 0, N % 8 == 0, tensors < 2 GiB (include/gd_nn.h gd_nn_gemm_supported)."""
    if not _gemm_ok(x, weight, bias, residual):
        return False
    N = weight.shape[0] // 2 if geglu else weight.shape[0]
    return bool(lib().gd_nn_gemm_supported(x.numel() // x.shape[-1], x.shape[-1], N, int(geglu)))


def gemm(x, weight, bias=None, residual=None):
    """``F.linear(x, weight, bias)``, or with ``residual`` ``torch.addmm(residual + bias, x, weight.T)`` (ONE rounding), on the own
    GEMM (csrc/nn_gemm.hip: persistent 256 x 256 x 64 tiles, ten-slot LDS-DMA ring), inference only."""
    K = weight.shape[1]
    M = x.numel() // K
    y = torch.empty(x.shape[:-1] + (weight.shape[0],), dtype=torch.bfloat16, device=x.device)
    p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_gemm_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), weight.data_ptr(), p(bias),
                                   p(residual), y.data_ptr(), M, K, weight.shape[0])
    if ret < 0:
        raise RuntimeError(f"gd_nn_gemm_forward failed ({ret}): {L.gd_nn_gemm_last_error().decode()}")
    return y


def gemm_geglu(x, weight, bias=None):
    """diffusers ``GEGLU(K, inner)``: ``h, g = F.linear(x, weight, bias).chunk(2, -1); h * gelu(g)`` as ONE kernel -- the GEGLU in
    the own GEMM's epilogue (the [M][2 inner] projection output is never written)."""
    K = weight.shape[1]
    M, inner = x.numel() // K, weight.shape[0] // 2
    y = torch.empty(x.shape[:-1] + (inner,), dtype=torch.bfloat16, device=x.device)
    L = lib()
    with torch.cuda.device(x.device):
        ret = L.gd_nn_gemm_geglu_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), weight.data_ptr(),
                                         None if bias is None else bias.data_ptr(), y.data_ptr(), M, K, inner)
    if ret < 0:
        raise RuntimeError(f"gd_nn_gemm_geglu_forward failed ({ret}): {L.gd_nn_gemm_last_error().decode()}")
    return y


def linear(x, weight, bias=None, residual=None):
    """``F.linear(x, weight, bias) + residual`` on the own MFMA implicit-GEMM kernel (one tap), inference only."""
    K = weight.shape[1]
    M = x.numel() // K
    y = torch.empty(x.shape[:-1] + (weight.shape[0],), dtype=torch.bfloat16, device=x.device)
    p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    with torch.cuda.device(x.device):
        ret = lib().gd_nn_linear_forward(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), weight.data_ptr(),
                                         p(bias), p(residual), y.data_ptr(), M, K, weight.shape[0])
    if ret < 0:
        raise RuntimeError(f"gd_nn_linear_forward failed ({ret}): {lib().gd_nn_conv_last_error().decode()}")
    return y
