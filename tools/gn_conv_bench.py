#!/usr/bin/env python
"""Fused GN+SiLU+conv3x3 (patch kernel) vs GN kernel + implicit-GEMM conv on the workload's shapes."""
import sys
import time
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

SH = [(8, 128, 128, 512), (8, 128, 256, 256), (8, 256, 256, 256), (8, 256, 512, 128), (8, 512, 512, 128), (8, 512, 512, 64),
      (16, 320, 320, 64), (16, 640, 640, 32), (16, 1280, 1280, 16), (16, 960, 320, 64), (16, 1920, 640, 32),
      (16, 2560, 1280, 16), (16, 640, 1280, 16), (16, 320, 640, 32)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for (N, ci, co, hw) in SH:
    cl = torch.channels_last
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=cl)
    b = torch.randn(co, device="cuda").to(torch.bfloat16)
    gw = torch.ones(ci, device="cuda", dtype=torch.bfloat16)
    gb = torch.zeros(ci, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * N * hw * hw * co * ci * 9
    with torch.no_grad():
        t_un = timeit(lambda: nn_ops.conv3x3(nn_ops.group_norm_silu(x, gw, gb, 32, 1e-5, True), w, b))
        t_conv = timeit(lambda: nn_ops.conv3x3(x, w, b))
        t_fu = timeit(lambda: nn_ops.gn_conv3x3(x, gw, gb, 32, 1e-5, True, w, b))
        err = (nn_ops.gn_conv3x3(x, gw, gb, 32, 1e-5, True, w, b).float()
               - nn_ops.conv3x3(nn_ops.group_norm_silu(x, gw, gb, 32, 1e-5, True), w, b).float()).abs().max().item()
    print(f"N{N} {ci:4d}->{co:4d} @{hw:3d}: gn+conv {t_un*1e6:7.1f}us (conv alone {t_conv*1e6:7.1f}us {fl/t_conv/1e12:5.0f}TF) | "
          f"fused {t_fu*1e6:7.1f}us  ({t_un/t_fu:4.2f}x)  e{err:.3f}")
