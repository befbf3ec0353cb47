#!/usr/bin/env python
"""Winograd F(2,3)-along-x convolution (csrc/nn_conv_wino.h) vs the direct patch-staged kernel: time + error vs fp32."""
import sys
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

SH = [(8, 128, 128, 512), (8, 128, 256, 256), (8, 256, 256, 256), (8, 256, 512, 128), (8, 512, 512, 128), (8, 512, 512, 64),
      (16, 320, 320, 64), (16, 640, 320, 64), (16, 960, 320, 64), (16, 640, 640, 32), (16, 1280, 1280, 32),
      (16, 1280, 1280, 16), (1, 128, 128, 512), (2, 320, 320, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SH = [(2, 64, 128, 48), (8, 128, 128, 512), (8, 256, 256, 256)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


ABL = len(sys.argv) > 1 and sys.argv[1] == "abl"
if ABL:
    SH = [(8, 128, 128, 512), (8, 256, 256, 256), (8, 512, 512, 128), (16, 320, 320, 64), (16, 1280, 1280, 32)]
torch.manual_seed(0)
for (N, ci, co, hw) in SH:
    cl = torch.channels_last
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=cl)
    b = torch.randn(co, device="cuda").to(torch.bfloat16)
    r = torch.randn(N, co, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    fl = 2.0 * N * hw * hw * co * ci * 9
    if ABL:
        with torch.no_grad():
            t_w = timeit(lambda: nn_ops._wino_launch(x, w, b, None, co), 20)
        print(f"   N{N} {ci:4d}->{co:4d} @{hw:3d}: wino {t_w*1e6:7.1f}us {fl/t_w/1e12:5.0f}TF", flush=True)
        continue
    with torch.no_grad():
        ref = F.conv2d(x.float(), w.float(), b.float(), padding=1) + r.float()
        yd = nn_ops._patch_launch(x, w, b, r, co)
        yw = nn_ops._wino_launch(x, w, b, r, co)
        sc = ref.abs().max().item()
        ed = (yd.float() - ref).abs().max().item() / sc
        ew = (yw.float() - ref).abs().max().item() / sc
        rd = ((yd.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        rw = ((yw.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        del ref
        t_p = timeit(lambda: nn_ops._patch_launch(x, w, b, r, co))
        t_w = timeit(lambda: nn_ops._wino_launch(x, w, b, r, co))
    gtxt = ""
    if N * hw * hw >= 65536 * 8 and ci % 32 == 0:      # GroupNorm-in-loader pair (the VAE / UNet ResnetBlock front half)
        with torch.no_grad():
            gw = torch.ones(ci, device="cuda", dtype=torch.bfloat16); gb = torch.zeros(ci, device="cuda", dtype=torch.bfloat16)
            mr = torch.tensor([0.0, 1.0], device="cuda").repeat(N * 32).contiguous()
            L = nn_ops.lib()
            yg = torch.empty_like(yw)
            st = torch.cuda.current_stream().cuda_stream
            fd = lambda: L.gd_nn_conv3x3_gn_forward(st, x.data_ptr(), mr.data_ptr(), gw.data_ptr(), gb.data_ptr(), 32, 1, w.data_ptr(),
                                                    b.data_ptr(), 0, r.data_ptr(), yg.data_ptr(), N, hw, hw, ci, co)
            fw = lambda: nn_ops._wino_gn_launch(x, mr, gw, gb, 32, True, w, b, r, co)
            t_gd, t_gw = timeit(fd), timeit(fw)
            gtxt = f" | GN: direct {t_gd*1e6:7.1f}us wino {t_gw*1e6:7.1f}us ({t_gd/t_gw:4.2f}x)"
    print(f"N{N} {ci:4d}->{co:4d} @{hw:3d}: direct {t_p*1e6:7.1f}us {fl/t_p/1e12:5.0f}TF | wino {t_w*1e6:7.1f}us {fl/t_w/1e12:5.0f}TF "
          f"({t_p/t_w:4.2f}x)  max err/max|ref| direct {ed:.2e} wino {ew:.2e}  rel rms direct {rd:.2e} wino {rw:.2e}" + gtxt, flush=True)
