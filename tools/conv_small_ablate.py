#!/usr/bin/env python
"""The split-K implicit-GEMM convolutions of the one-view-per-GPU step (UNet at 2 latents): time per launch under
whatever libgd_nn the env selects (GD_NN_LIB=ablate/... built with -DGD_CONV_ABLATE=n; timing only)."""
import os
import sys
import torch
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
from garmentdreamer_amd import nn_ops

SH = [(2, 320, 320, 64), (2, 640, 640, 32), (2, 1280, 1280, 16), (2, 1280, 1280, 8), (2, 2560, 1280, 16), (2, 960, 640, 32)]
out = []
for (N, ci, co, hw) in SH:
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(5):
            nn_ops.conv3x3(x, w, b)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                nn_ops.conv3x3(x, w, b)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 100 * 1e-3
    out.append(f"{ci}->{co}@{hw}: {t*1e6:5.1f}us {2.0*N*hw*hw*co*ci*9/t/1e12:4.0f}TF")
print(f"{os.path.basename(os.environ.get('GD_NN_LIB', 'default')):28s}", " | ".join(out))
