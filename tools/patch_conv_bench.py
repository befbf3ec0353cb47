#!/usr/bin/env python
"""Plain conv3x3: implicit-GEMM kernel vs the patch-staged kernel with an LDS-DMA patch."""
import sys
import time
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

SH = [(8, 128, 128, 512), (8, 128, 256, 256), (8, 256, 128, 256), (8, 256, 256, 256), (8, 256, 512, 128), (8, 512, 256, 128),
      (8, 512, 512, 128), (8, 512, 512, 64), (16, 320, 320, 64), (16, 640, 320, 64), (16, 960, 320, 64), (16, 640, 640, 64),
      (16, 640, 640, 32), (16, 1280, 1280, 32), (16, 1280, 1280, 16), (16, 2560, 1280, 16), (16, 1920, 1280, 32), (16, 640, 1280, 16), (1, 128, 128, 512), (1, 256, 256, 256),
      (1, 512, 512, 128), (2, 1280, 1280, 32)]


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
    fl = 2.0 * N * hw * hw * co * ci * 9
    with torch.no_grad():
        t_i = timeit(lambda: nn_ops._conv_launch(x, w, b, None, co))
        t_p = timeit(lambda: nn_ops._patch_launch(x, w, b, None, co))
        err = (nn_ops._conv_launch(x, w, b, None, co).float() - nn_ops._patch_launch(x, w, b, None, co).float()).abs().max().item()
    print(f"N{N} {ci:4d}->{co:4d} @{hw:3d}: implicit {t_i*1e6:7.1f}us {fl/t_i/1e12:5.0f}TF | patch {t_p*1e6:7.1f}us {fl/t_p/1e12:5.0f}TF "
          f"({t_i/t_p:4.2f}x) e{err:.4f}")
