#!/usr/bin/env python
"""Split-K sweep of the implicit-GEMM 3x3 convolution on the small-map shapes of the UNet at batch 2 / 4
(one or two views per GPU, the VSD iteration).  Device time per call from a hipGraph of 20 calls.
    python tools/splitk_sweep.py [batch]"""
import sys
import torch
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
import garmentdreamer_amd  # noqa: F401
from garmentdreamer_amd import nn_ops
from garmentdreamer_amd.nn_ops import conv3x3

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
MID = len(sys.argv) > 2 and sys.argv[2] == "mid"   # the 256..511-tile layers of two views per GPU
SHAPES = [(320, 320, 64), (640, 320, 64), (960, 320, 64), (320, 640, 32), (640, 640, 32), (1280, 640, 32), (960, 640, 32),
          (1920, 640, 32), (640, 1280, 16), (1280, 1280, 16), (2560, 1280, 16), (1920, 1280, 16), (1280, 1280, 8),
          (2560, 1280, 8)]
if MID:
    SHAPES = [(512, 512, 64), (1280, 1280, 32), (320, 320, 64), (640, 320, 64), (960, 320, 64), (2560, 1280, 32), (640, 640, 64)]
L = nn_ops.lib()


def graph_time(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


for ci, co, hw in SHAPES:
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device="cuda").to(torch.bfloat16)
    steps = 9 * ci // 64
    tiles = ((N * hw * hw + 127) // 128) * ((co + 127) // 128)
    out = []
    with torch.no_grad():
        L.gd_nn_conv_force_split(-1)
        ref = conv3x3(x, w, b).float()
        t_auto = graph_time(lambda: conv3x3(x, w, b))
        for S in (1, 2, 3, 4, 5, 6, 8, 9, 12, 15, 18, 24, 30, 36, 45):
            if S > steps // 2:
                continue
            L.gd_nn_conv_force_split(S)
            t = graph_time(lambda: conv3x3(x, w, b))
            err = (conv3x3(x, w, b).float() - ref).abs().max().item()
            out.append((t, S, err))
        L.gd_nn_conv_force_split(-1)
    best = min(out)
    print(f"N{N} {ci:4d}->{co:4d} @{hw:2d} tiles {tiles:4d} steps {steps:3d}: auto {t_auto:6.1f} us | best S={best[1]:2d} {best[0]:6.1f} us | "
          + " ".join(f"S{S}:{t:.1f}" for t, S, e in out) + f" | maxerr {max(e for _, _, e in out):.3f}", flush=True)
