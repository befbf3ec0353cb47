#!/usr/bin/env python
"""Time the conv kernel on a few shapes with whatever libgd_nn the env selects (GD_NN_LIB=ablate/...):
bottleneck ablations built with -DGD_CONV_ABLATE=n (results are wrong by construction, timing only)."""
import os
import sys
import time
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

SH = [(8, 512, 512, 128, 2), (8, 256, 256, 256, 2), (8, 128, 128, 512, 1), (16, 1280, 1280, 16, 0), (16, 640, 640, 32, 0),
      (16, 320, 320, 64, 0)]
L = nn_ops.lib()
out = []
for (N, ci, co, hw, v) in SH:
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda").to(torch.bfloat16)
    L.gd_nn_conv_force_variant(v)
    with torch.no_grad():
        for _ in range(5):
            nn_ops.conv3x3(x, w, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            nn_ops.conv3x3(x, w, b)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20
    out.append(f"{ci}->{co}@{hw} v{v}: {t*1e6:6.1f}us {2.0*N*hw*hw*co*ci*9/t/1e12:6.0f}TF")
print(os.environ.get("GD_NN_LIB", "default"), " | ".join(out))
