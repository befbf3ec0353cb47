#!/usr/bin/env python
"""Per-shape decision table for routing stride-1 3x3 convolutions to the Winograd kernel: the product's current choice
(nn_ops._conv_launch with GD_NN_WINO=0: patch-staged / implicit-GEMM / split-K) against nn_ops._wino_launch, every
stride-1 shape of the SDS step (UNet at batch 2V, VAE encoder + its input gradients at batch V).  Usage: [V]"""
import os
import sys
import time
os.environ["GD_NN_WINO"] = "0"
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = 2 * V
# (N, Cin, Cout, HW, launches per step, per-image bias, residual)
SH = [(V, 128, 128, 512, 8, 0, 1), (V, 128, 256, 256, 1, 0, 0), (V, 256, 128, 256, 1, 0, 0), (V, 256, 256, 256, 6, 0, 1),
      (V, 256, 512, 128, 1, 0, 0), (V, 512, 256, 128, 1, 0, 0), (V, 512, 512, 128, 6, 0, 1), (V, 512, 512, 64, 16, 0, 1),
      (B, 320, 320, 64, 7, 1, 1), (B, 640, 320, 64, 2, 1, 0), (B, 960, 320, 64, 1, 1, 0), (B, 640, 640, 64, 1, 0, 0),
      (B, 320, 640, 32, 1, 1, 0), (B, 640, 640, 32, 6, 1, 1), (B, 1920, 640, 32, 1, 1, 0), (B, 1280, 640, 32, 1, 1, 0),
      (B, 960, 640, 32, 1, 1, 0), (B, 1280, 1280, 32, 1, 0, 0), (B, 640, 1280, 16, 1, 1, 0), (B, 1280, 1280, 16, 7, 1, 1),
      (B, 2560, 1280, 16, 2, 1, 0), (B, 1920, 1280, 16, 1, 1, 0)]


def timeit(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


torch.manual_seed(0)
tot_p = tot_best = 0.0
for (N, ci, co, hw, cnt, pib, res) in SH:
    cl = torch.channels_last
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=cl)
    b = torch.randn(N, co, device="cuda").to(torch.bfloat16) if pib else torch.randn(co, device="cuda").to(torch.bfloat16)
    r = torch.randn(N, co, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl) if res else None
    with torch.no_grad():
        t_p = timeit(lambda: nn_ops._conv_launch(x, w, b, r, co))
        t_w = timeit(lambda: nn_ops._wino_launch(x, w, b, r, co))
        t_d = timeit(lambda: nn_ops._wide_launch(x, w, b, r, co))
        t_p2 = timeit(lambda: nn_ops._conv_launch(x, w, b, r, co))
        t_w2 = timeit(lambda: nn_ops._wino_launch(x, w, b, r, co))
        t_d2 = timeit(lambda: nn_ops._wide_launch(x, w, b, r, co))
    t_p, t_w, t_d = min(t_p, t_p2), min(t_w, t_w2), min(t_d, t_d2)
    fl = 2.0 * N * hw * hw * co * ci * 9
    tot_p += cnt * t_p
    tot_best += cnt * min(t_p, t_w, t_d)
    best = min((t_p, ""), (t_w / 0.97, "WINO"), (t_d / 0.97, "WIDE"))[1]
    print(f"N{N:2d} {ci:4d}->{co:4d} @{hw:3d} x{cnt:2d}: product {t_p*1e6:7.1f}us {fl/t_p/1e12:5.0f}TF | wino {t_w*1e6:7.1f}us {fl/t_w/1e12:5.0f}TF "
          f"{t_p/t_w:4.2f}x | wide {t_d*1e6:7.1f}us {fl/t_d/1e12:5.0f}TF {t_p/t_d:4.2f}x  {best}", flush=True)
print(f"per step: product {tot_p*1e3:.2f} ms, best-of {tot_best*1e3:.2f} ms")
