#!/usr/bin/env python
"""dL/d(rgb) of the SDS guidance (full bf16 SD-2.1, hipGraphs) for 8 views vs the first 4 of them alone, with and
without batch-invariant kernel selection (nn_ops.set_route_scale)."""
import sys

import torch

sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
from garmentdreamer_amd import nn_ops  # noqa: E402
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance  # noqa: E402

V = 8
dev = torch.device("cuda", 0)
graphs = "--eager" not in sys.argv
g = torch.Generator().manual_seed(3)
rgb = torch.rand(V, 512, 512, 3, generator=g).to(dev)
noise = torch.randn(V, 4, 64, 64, generator=g).to(dev)
vnoise = torch.randn(V, 4, 64, 64, generator=g).to(dev)
ts = torch.randint(20, 981, (V,), generator=g).to(dev)
el = torch.linspace(-10, 40, V).to(dev)
az = torch.linspace(-180, 135, V).to(dev)
cd = torch.full((V,), 3.0, device=dev)
pe = PromptEmbeddings.random(dev)


def sel(n, rank=0):
    return slice(None) if n == V else slice(rank, V, V // n)      # dist.shard_views: rank r holds views r, r + k, ...


def run(n, scale, rank=0):
    nn_ops.set_route_scale(scale, rank if scale > 1 else 0)
    guidance = StableDiffusionGuidance({"guidance_scale": 7.5, "grad_clip": [0, 1.5, 2.0, 1000], "use_hip_graphs": graphs},
                                       device=dev)
    guidance.update_step(0, 10)
    i = sel(n, rank)
    x = rgb[i].clone().requires_grad_(True)
    out = guidance(x, pe, el[i], az[i], cd[i], noise=noise[i], timesteps=ts[i], vae_noise=vnoise[i])
    (out["loss_sds"] * n).backward()          # undo the 1 / batch_size
    torch.cuda.synchronize()
    nn_ops.set_route_scale(1)
    return x.grad.clone(), float(out["loss_sds"]) * n


g8, l8 = run(8, 1)
g8b, _ = run(8, 1)
print(f"8 views twice: max|d| {(g8 - g8b).abs().max().item():.3e}")
for scale, rank in ((1, 0), (1, 1), (2, 0), (2, 1)):
    g4, l4 = run(4, scale, rank)
    ref = g8[sel(4, rank)]
    d = (g4 - ref).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(g4.flatten().double(), ref.flatten().double(), dim=0).item()
    print(f"views {rank}, {rank + 2}, .. of 8 alone, route scale {scale}: dL/drgb max|d| {d:.3e} of {g8.abs().max().item():.3e}, cos {cos:.10f}")

