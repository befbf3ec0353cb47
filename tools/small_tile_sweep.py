#!/usr/bin/env python
"""Small-M layers (one view per GPU: UNet at batch 2, VAE at batch 1; the VSD iteration: batch 1): the implicit-GEMM kernel's
tile variants WITHOUT split-K against today's route (split-K + reduce launch) and, for the linears, against hipBLASLt.
Device time per call from a hipGraph of 20 calls.   python tools/small_tile_sweep.py [conv|lin] [batch]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
import garmentdreamer_amd  # noqa: F401
from garmentdreamer_amd import nn_ops

what = sys.argv[1] if len(sys.argv) > 1 else "conv"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L = nn_ops.lib()
nn_ops._WINO = False
VARIANTS = {0: "128x128", 5: "64x128", 6: "64x64", 7: "128x64"}


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


def force(v, s):
    L.gd_nn_conv_force_variant(v)
    L.gd_nn_conv_force_split(s)


if what == "conv":
    SHAPES = [(N, 320, 320, 64), (N, 640, 320, 64), (N, 960, 320, 64), (N, 320, 640, 32), (N, 640, 640, 32), (N, 1280, 640, 32),
              (N, 960, 640, 32), (N, 1920, 640, 32), (N, 1280, 1280, 32), (N, 640, 1280, 16), (N, 1280, 1280, 16), (N, 2560, 1280, 16),
              (N, 1920, 1280, 16), (N, 1280, 1280, 8), (N, 2560, 1280, 8), (max(N // 2, 1), 512, 512, 64), (max(N // 2, 1), 512, 512, 128),
              (max(N // 2, 1), 256, 256, 256)]
    for n, ci, co, hw in SHAPES:
        x = torch.randn(n, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device="cuda").to(torch.bfloat16)
        fl = 2.0 * n * hw * hw * co * 9 * ci
        with torch.no_grad():
            force(-1, -1)
            ref = nn_ops._conv_launch(x, w, b, None, co).float()
            t_auto = graph_time(lambda: nn_ops._conv_launch(x, w, b, None, co))
            row = [f"auto {t_auto:6.1f}us {fl / t_auto / 1e6:5.0f}TF"]
            best = (t_auto, "auto")
            for v, name in VARIANTS.items():
                force(v, 1)
                t = graph_time(lambda: nn_ops._conv_launch(x, w, b, None, co))
                err = (nn_ops._conv_launch(x, w, b, None, co).float() - ref).abs().max().item()
                row.append(f"{name} {t:6.1f}us" + ("" if err < 0.05 else f" ERR{err:.2g}"))
                best = min(best, (t, name))
            force(-1, -1)
        print(f"N{n} {ci:4d}->{co:4d} @{hw:3d}: " + " | ".join(row) + f"  BEST {best[1]} {t_auto / best[0]:.2f}x", flush=True)
else:
    levels = [(320, 4096), (640, 1024), (1280, 256), (1280, 64)]
    if N >= 8:      # the 8-view step: the 256 x 256 and 128 x 256 tiles belong in the comparison, TF/s in the table
        VARIANTS = {0: "128x128", 1: "128chx256px", 2: "256x256", 5: "64x128", 7: "128x64"}
    shapes = [(N * tok, K, Nn) for C, tok in levels for K, Nn in ((C, C), (C, 2 * C), (C, 8 * C), (4 * C, C))]
    shapes += [(N * 77, 1024, 24960)]                         # every cross-attention's K | V of the text tokens, one product
    if N >= 8:
        shapes += [(N // 2 * 4096, 512, 1536), (N // 2 * 4096, 512, 512), (N // 2 * 16384, 256, 512)]   # VAE: qkv, to_out, 1x1 shortcut
    for Mx, K, Nn in shapes:
        if True:
            x = torch.randn(Mx, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(Nn, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(Nn, device="cuda").to(torch.bfloat16)
            with torch.no_grad():
                ref = F.linear(x, w, b).float()
                t0 = graph_time(lambda: F.linear(x, w, b))
                row = [f"hipBLASLt {t0:6.1f}us {2.0 * Mx * K * Nn / t0 / 1e6:5.0f}TF"]
                best = (t0, "lib")
                for v, name in VARIANTS.items():
                    force(v, 1)
                    t = graph_time(lambda: nn_ops.linear(x, w, b))
                    err = (nn_ops.linear(x, w, b).float() - ref).abs().max().item()
                    row.append(f"{name} {t:6.1f}us" + ("" if err < 0.05 * ref.abs().max().item() + 0.02 else f" ERR{err:.2g}"))
                    best = min(best, (t, name))
                force(-1, -1)
            print(f"M{Mx:6d} K{K:5d} N{Nn:6d}: " + " | ".join(row) + f"  BEST {best[1]} {t0 / best[0]:.2f}x", flush=True)
