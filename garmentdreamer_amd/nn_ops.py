"""Torch-facing wrappers of ``libgd_nn.so`` (C-ABI: include/gd_nn.h) -- hand-written gfx950 kernels for
the guidance step.  On HIP tensors the kernels are mandatory (missing library -> error, no silent
eager fallback); on CPU tensors (tests, cpu_baseline) the plain PyTorch ops of the same math run.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgd_nn.so")
_lib = None

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "gd_nn_groupnorm_silu_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "gd_nn_groupnorm_silu_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gd_nn_groupnorm_ws_bytes": (C.c_size_t, [_i, _i]),
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
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _check(ret, what):
    if ret < 0:
        raise RuntimeError(f"{what} failed ({ret}): {lib().gd_nn_last_error().decode()}")


def _is_nhwc_bf16(x):
    return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)


class _GroupNormSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu):
        N, Cc, H, W = x.shape
        L = lib()
        y = torch.empty_like(x, memory_format=torch.channels_last)
        ws = torch.empty(N * groups * 2, dtype=torch.float64, device=x.device)
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
        ws = torch.empty(N * ctx.groups * 2, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            _check(L.gd_nn_groupnorm_silu_backward(stream, x.data_ptr(), dy.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                   mr.data_ptr(), dx.data_ptr(), N, H * W, Cc, ctx.groups,
                                                   int(ctx.silu), ws.data_ptr()), "gd_nn_groupnorm_silu_backward")
        return dx, None, None, None, None, None


def group_norm_silu(x, weight, bias, groups: int, eps: float, silu: bool = True):
    """``silu(group_norm(x))`` (or just group_norm).  HIP kernel for bf16 NHWC tensors on the GPU."""
    if x.is_cuda:
        if weight.requires_grad or bias.requires_grad:
            raise RuntimeError("group_norm_silu HIP kernel computes input gradients only (frozen weights)")
        if x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0:
            if not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
            return _GroupNormSiLU.apply(x, weight, bias, groups, eps, silu)
        # fp32 GPU runs (parity checks of the bf16 path) use torch's ops
    y = F.group_norm(x, groups, weight, bias, eps)
    return F.silu(y) if silu else y
