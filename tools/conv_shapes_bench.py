#!/usr/bin/env python
"""Time every distinct conv2d / linear shape of the SD-2.1 UNet (batch 2V) and VAE encoder (batch V)
through PyTorch-ROCm (MIOpen / hipBLASLt), forward and (VAE) input-gradient, to see where a hand-written
MFMA kernel would pay.  Usage: python tools/conv_shapes_bench.py [V]"""
import collections
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd.guidance import sd21  # noqa

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
shapes = collections.OrderedDict()


def hook(name, kind):
    def f(mod, inp, out):
        x = inp[0]
        if isinstance(mod, nn.Conv2d):
            key = (kind, "conv", tuple(x.shape), mod.out_channels, mod.kernel_size[0], mod.stride[0], mod.padding[0])
        else:
            key = (kind, "linear", tuple(x.shape), mod.out_features)
        shapes.setdefault(key, 0)
        shapes[key] += 1
    return f


with torch.device("meta"):
    unet = sd21.UNet2DConditionModel()
    vae = sd21.AutoencoderKLEncoder()
for m in unet.modules():
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        m.register_forward_hook(hook("", "unet"))
for m in vae.modules():
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        m.register_forward_hook(hook("", "vae"))
with torch.device("meta"):
    unet(torch.zeros(2 * V, 4, 64, 64), torch.zeros(2 * V), torch.zeros(2 * V, 77, 1024))
    vae.encode(torch.zeros(V, 3, 512, 512))


def timeit(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


rows = []
for key, cnt in shapes.items():
    kind, op = key[0], key[1]
    if op == "conv":
        _, _, xs, co, k, s, p = key
        x = torch.randn(xs, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        conv = nn.Conv2d(xs[1], co, k, stride=s, padding=p).to(dev, torch.bfloat16).to(memory_format=torch.channels_last)
        for q in conv.parameters():
            q.requires_grad_(False)
        with torch.no_grad():
            t_f = timeit(lambda: conv(x))
        ho = (xs[2] + 2 * p - k) // s + 1
        fl = 2.0 * xs[0] * ho * ho * co * xs[1] * k * k
        t_b = 0.0
        if kind == "vae":
            xg = x.clone().requires_grad_(True)
            y = conv(xg)
            g = torch.randn_like(y)
            t_b = timeit(lambda: torch.autograd.grad(y, xg, g, retain_graph=True))
        rows.append((kind, f"conv{k}x{k}s{s} {xs[1]}->{co} @{xs[2]} N{xs[0]}", cnt, fl, t_f, t_b))
    else:
        _, _, xs, co = key
        x = torch.randn(xs, device=dev, dtype=torch.bfloat16)
        lin = nn.Linear(xs[-1], co).to(dev, torch.bfloat16)
        with torch.no_grad():
            t_f = timeit(lambda: lin(x))
        fl = 2.0 * x.numel() * co
        rows.append((kind, f"linear {xs[-1]}->{co} M{x.numel() // xs[-1]}", cnt, fl, t_f, 0.0))

tot_f = tot_b = 0.0
print(f"{'net':5s} {'op':44s} {'cnt':>3s} {'GFLOP':>8s} {'fwd us':>8s} {'TF/s':>7s} {'bwd us':>8s} {'TF/s':>7s} {'tot ms':>7s}")
for kind, name, cnt, fl, tf, tb in sorted(rows, key=lambda r: -(r[4] + r[5]) * r[2]):
    tot = cnt * (tf + tb) * 1e3
    tot_f += cnt * tf
    tot_b += cnt * tb
    print(f"{kind:5s} {name:44s} {cnt:3d} {fl / 1e9:8.1f} {tf * 1e6:8.1f} {fl / tf / 1e12:7.1f} "
          f"{tb * 1e6:8.1f} {(fl / tb / 1e12 if tb else 0):7.1f} {tot:7.2f}")
print(f"total fwd {tot_f * 1e3:.2f} ms, total vae dgrad {tot_b * 1e3:.2f} ms")
