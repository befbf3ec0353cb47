"""Full-size SDS loop across densify/prune events: 100k Gaussians x 8 views @512^2, SD-2.1-size nets with hipGraphs,
global steps 395..620 (events at 400, 500, 600).  Prints P over time, step times before / after, and health."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd  # noqa
import torch
from garmentdreamer_amd import cameras as gcam
from garmentdreamer_amd.gaussian_model import GaussianModel
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance
from garmentdreamer_amd.scene import synthetic_gaussians
from garmentdreamer_amd.sds_loop import SDSLoop

dev = torch.device("cuda", 0)
V, P = 8, 100000
guidance = StableDiffusionGuidance({"guidance_scale": 100.0, "grad_clip": [0, 1.5, 2.0, 1000], "use_hip_graphs": True}, device=dev)
gm = GaussianModel.from_activated(synthetic_gaussians(P, seed=0, sh_degree=0), sh_degree=0, device=dev)
loop = SDSLoop(gm, guidance, PromptEmbeddings.random(dev), torch.ones(3, device=dev), densify_seed=5)
loop.global_step = 395
gen = torch.Generator(device=dev).manual_seed(1)
times, Ps = [], []
for s in range(226):
    batch = gcam.orbit_batch(V, elevation_deg=15.0, camera_distance=2.75, fovy_deg=55.0, height=512, width=512, azimuth_offset_deg=3.0 * s)
    noise = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    vn = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    t = torch.randint(20, 981, (V,), device=dev, generator=gen)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = loop.step(batch, noise=noise, timesteps=t, vae_noise=vn)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    Ps.append(gm.get_xyz.shape[0])
    if out["densified"]:
        print(f"step {loop.global_step - 1}: densify/prune -> P = {Ps[-1]}  ({times[-1]*1e3:.1f} ms)", flush=True)
fin = all(bool(torch.isfinite(p).all()) for p in gm.parameters())
print("P first/last", Ps[0], Ps[-1], "finite", fin)
print("ms/step median: steps 1-4 %.2f | after 1st event %.2f | after 3rd event %.2f" % (
    1e3 * sorted(times[1:5])[2], 1e3 * sorted(times[10:100])[45], 1e3 * sorted(times[210:226])[8]))
assert fin
