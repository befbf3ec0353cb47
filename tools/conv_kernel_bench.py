#!/usr/bin/env python
"""Own MFMA conv3x3 vs MIOpen on the dominant shapes.  python tools/conv_kernel_bench.py"""
import sys
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd.nn_ops import conv3x3

# (N, Cin, Cout, HW, launches per SDS step at V=8: UNet batch 16 forward; VAE batch 8 forward + dgrad)
SHAPES = [(8, 128, 128, 512, 8), (8, 128, 256, 256, 1), (8, 256, 128, 256, 1), (8, 256, 256, 256, 6),
          (8, 256, 512, 128, 1), (8, 512, 256, 128, 1), (8, 512, 512, 128, 6), (8, 512, 512, 64, 16),
          (16, 320, 320, 64, 7), (16, 960, 320, 64, 1), (16, 640, 320, 64, 2), (16, 640, 640, 64, 1),
          (16, 320, 640, 32, 1), (16, 640, 640, 32, 6), (16, 1920, 640, 32, 1), (16, 1280, 640, 32, 1),
          (16, 960, 640, 32, 1), (16, 1280, 1280, 32, 1),
          (16, 640, 1280, 16, 1), (16, 1280, 1280, 16, 7), (16, 2560, 1280, 16, 2), (16, 1920, 1280, 16, 1),
          (16, 1280, 1280, 8, 11), (16, 2560, 1280, 8, 3)]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


tot = {}
for (N, ci, co, hw, cnt) in SHAPES:
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda").to(torch.bfloat16)
    fl = 2.0 * N * hw * hw * co * ci * 9
    from garmentdreamer_amd import nn_ops
    L = nn_ops.lib()
    with torch.no_grad():
        ref = F.conv2d(x, w, b, padding=1).float()
        res = []
        for v in (0, 1, 2, -1):
            L.gd_nn_conv_force_variant(v)
            t = timeit(lambda: conv3x3(x, w, b))
            err = (conv3x3(x, w, b).float() - ref).abs().max().item()
            res.append(f"v{v}: {t*1e6:7.1f}us {fl/t/1e12:6.1f}TF e{err:.2f}")
        L.gd_nn_conv_force_variant(-1)
        t_mio = timeit(lambda: F.conv2d(x, w, b, padding=1))
    tot["flops"] = tot.get("flops", 0.0) + fl * cnt
    tot["auto"] = tot.get("auto", 0.0) + t * cnt
    tot["miopen"] = tot.get("miopen", 0.0) + t_mio * cnt
    print(f"N{N} {ci:4d}->{co:4d} @{hw:3d} x{cnt:2d} ({t*cnt*1e3:5.2f} ms/step): " + " | ".join(res) + f" | miopen {t_mio*1e6:7.1f}us {fl/t_mio/1e12:6.1f}TF")
print(f"per SDS step: {tot['flops']/1e12:.2f} TFLOP, own {tot['auto']*1e3:.2f} ms ({tot['flops']/tot['auto']/1e12:.0f} TF/s), "
      f"miopen {tot['miopen']*1e3:.2f} ms ({tot['flops']/tot['miopen']/1e12:.0f} TF/s)")
