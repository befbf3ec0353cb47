#!/usr/bin/env python
"""Which operators of the SD-2.1 UNet forward are NOT batch-invariant on this box?

Runs the full-size bf16 UNet on a batch of B latents; every operator call on the way (own HIP entry points and the
library GEMMs behind F.linear / torch.matmul) is repeated on the first B/2 samples of its own inputs -- with the
kernel selection told the whole batch is still B (nn_ops.set_route_scale(2)) and without -- and compared BIT-exactly
with the first half of the full-batch result.  Prints one line per (operator, shape) that differs.

  python tools/batch_invariance_probe.py [B]          # default 16 = 8 views x (text, uncond)
"""
import collections
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
from garmentdreamer_amd import nn_ops  # noqa: E402
from garmentdreamer_amd.guidance import sd21  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
stats = collections.OrderedDict()
depth = [0]


def half(v):
    if torch.is_tensor(v) and not isinstance(v, torch.nn.Parameter) and v.dim() >= 1 and v.shape[0] == B:
        return v[:B // 2]
    if isinstance(v, (tuple, list)):
        return type(v)(half(t) for t in v)
    return v


def first(o):
    return o[0] if isinstance(o, (tuple, list)) else o


def wrap(fn, name):
    def w(*a, **k):
        out = fn(*a, **k)
        if depth[0]:
            return out
        o = first(out)
        if not (torch.is_tensor(o) and o.dim() >= 1 and o.shape[0] == B):
            return out
        depth[0] += 1
        try:
            a2, k2 = half(a), {kk: half(vv) for kk, vv in k.items()}
            res = {}
            for scale in (1, 2):
                nn_ops.set_route_scale(scale)
                o2 = first(fn(*a2, **k2))
                d = (o2.float() - o[:B // 2].float()).abs().max().item()
                res[scale] = d
            nn_ops.set_route_scale(1)
            res[0] = (first(fn(*a, **k)).float() - o.float()).abs().max().item()      # the same call again
        finally:
            depth[0] -= 1
        shapes = tuple(tuple(t.shape) for t in a if torch.is_tensor(t))[:2]
        key = (name, shapes)
        s = stats.setdefault(key, [0, 0.0, 0.0, float(o.detach().float().abs().max()), 0.0])
        s[0] += 1
        s[1] = max(s[1], res[1])
        s[2] = max(s[2], res[2])
        s[4] = max(s[4], res[0])
        return out
    return w


for n in ("linear_320", "conv3x3", "group_norm_silu", "attention_d64", "attention_d64_vt", "gn_conv3x3", "conv1x1",
          "conv3x3_s2", "conv3x3_small_cin", "add_layer_norm", "geglu", "resnet_block_frozen", "upsample2x_conv3x3",
          "linear_320_geglu"):
    if hasattr(sd21, n):
        setattr(sd21, n, wrap(getattr(sd21, n), n))
sd21._lib_linear = wrap(sd21._lib_linear, "library linear")
sd21._lib_addmm = wrap(sd21._lib_addmm, "library addmm")
torch.matmul = wrap(torch.matmul, "torch.matmul")
torch.bmm = wrap(torch.bmm, "torch.bmm")
torch.softmax = wrap(torch.softmax, "torch.softmax")

with torch.device(dev):
    unet = sd21.init_random_(sd21.UNet2DConditionModel()).to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(False)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 4, 64, 64, generator=g).to(dev, torch.bfloat16)
t = torch.randint(20, 980, (B,), generator=g).to(dev)
ctx = torch.randn(B, 77, 1024, generator=g).to(dev, torch.bfloat16)
with torch.no_grad():
    unet(x, t, ctx)          # the probing pass (operators with in-place residual updates are re-run: its result is not used)
    depth[0] = 1
    nn_ops.set_route_scale(1)
    y = unet(x, t, ctx)
    print(f"UNet run-to-run (same call twice): max|d| {(unet(x, t, ctx).float() - y.float()).abs().max().item():.3e}")
    y1 = unet(x[:B // 2], t[:B // 2], ctx[:B // 2])
    nn_ops.set_route_scale(2)
    y2 = unet(x[:B // 2], t[:B // 2], ctx[:B // 2])
    nn_ops.set_route_scale(1)
torch.cuda.synchronize()
print(f"whole UNet, first {B // 2} of {B} vs alone: max|d| tuned-per-batch {(y1.float() - y[:B // 2].float()).abs().max().item():.3e}, "
      f"batch-invariant selection {(y2.float() - y[:B // 2].float()).abs().max().item():.3e}  (|y| max {y.float().abs().max().item():.3e})")
HDR = f"{'operator':22s} {'calls':>5s} {'max|d| scale1':>14s} {'max|d| scale2':>14s} {'run-to-run':>11s} {'|out| max':>10s}  shapes"


def table():
    nbad = 0
    print(HDR)
    for (name, shapes), (cnt, d1, d2, om, rr) in stats.items():
        if d1 or d2 or rr:
            nbad += 1
            print(f"{name:22s} {cnt:5d} {d1:14.3e} {d2:14.3e} {rr:11.3e} {om:10.3e}  {shapes}")
    print(f"{nbad} of {len(stats)} (operator, shape) classes differ")

table()

# VAE encoder forward + backward (the SDS gradient's way back to the image), whole-network check
del unet
V = B // 2
with torch.device(dev):
    vae = sd21.init_random_(sd21.AutoencoderKLEncoder(), 1).to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(False)
img = torch.rand(V, 3, 512, 512, generator=g).to(dev, torch.bfloat16)
wgt = torch.randn(V, 4, 64, 64, generator=g).to(dev)
stats.clear()
B = V          # the wrappers halve along a leading dimension of V views now
depth[0] = 0
with torch.no_grad():
    vae.encode(img)
torch.cuda.synchronize()
print("VAE encoder forward (no grad), operator by operator:")
table()
stats.clear()
vae.encode(img.clone().requires_grad_(True))
torch.cuda.synchronize()
print("VAE encoder forward (autograd path, as the SDS step runs it), operator by operator:")
table()
depth[0] = 1


def vae_grad(n, scale):
    nn_ops.set_route_scale(scale)
    x = img[:n].clone().requires_grad_(True)
    lat = vae.encode(x).latent_dist.mean
    (lat.float() * wgt[:n]).sum().backward()
    nn_ops.set_route_scale(1)
    return lat.detach().float(), x.grad.float()


lat, gr = vae_grad(V, 1)
lat_b, gr_b = vae_grad(V, 1)
print(f"VAE run-to-run (same call twice): latent max|d| {(lat_b - lat).abs().max().item():.3e}, image-gradient max|d| {(gr_b - gr).abs().max().item():.3e}")
for scale in (1, 2):
    l2, g2 = vae_grad(V // 2, scale)
    print(f"VAE encode, first {V // 2} of {V} views alone, route scale {scale}: latent max|d| {(l2 - lat[:V // 2]).abs().max().item():.3e} "
          f"(|lat| {lat.abs().max().item():.2e}), image-gradient max|d| {(g2 - gr[:V // 2]).abs().max().item():.3e} (|g| {gr.abs().max().item():.2e})")
