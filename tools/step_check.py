#!/usr/bin/env python
"""Run a few SDS iterations of the benchmark workload and print per-step health: loss, gradient norm, visible
Gaussians, finiteness of the parameters (the timed bench does not look at values)."""
import argparse
import sys
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
import bench  # noqa: E402
from garmentdreamer_amd.gaussian_model import GaussianModel  # noqa: E402
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance  # noqa: E402
from garmentdreamer_amd.scene import synthetic_gaussians  # noqa: E402
from garmentdreamer_amd.sds_loop import SDSLoop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--no-graphs", action="store_true")
a = ap.parse_args()
args = argparse.Namespace(views=a.views, gaussians=100000, res=512)
dev = torch.device("cuda", 0)
g = GaussianModel.from_activated(synthetic_gaussians(100000, seed=0), device=dev)
guid = StableDiffusionGuidance({"guidance_scale": 100.0, "grad_clip": [0, 1.5, 2.0, 1000], "use_hip_graphs": not a.no_graphs},
                               device=dev)
loop = SDSLoop(g, guid, PromptEmbeddings.random(dev), torch.ones(3, device=dev))
gen = torch.Generator(device=dev)
for s in range(a.steps):
    batch = bench.camera_batch(args, s, list(range(a.views)))
    gen.manual_seed(1234 + 1000 * s)
    V = a.views
    noise = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    vn = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    t = torch.randint(20, 981, (V,), device=dev, generator=gen)
    out = loop.step(batch, noise=noise, timesteps=t, vae_noise=vn)
    print(f"step {s}: loss {out['loss'].item():.4f} sds {out['loss_sds'].item():.4f} grad_norm {float(out['grad_norm']):.4f} "
          f"visible {int(out['num_visible'])} params finite {bool(torch.isfinite(g._flat).all())} "
          f"|grad| {g.flat_grad.abs().max().item():.3e} xyz range {g._xyz.abs().max().item():.3f}")
