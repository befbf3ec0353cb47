#!/usr/bin/env python
"""Reproduces the hipGraph replay corruption that garmentdreamer_amd/_runtime_env.py works around.

A process that has already run a small guidance instance and one render call captures the full-size UNet / VAE
graphs; with ROCm's graph packet capture on (DEBUG_CLR_GRAPH_PACKET_CAPTURE unset or 1) the replayed VAE backward
returns NaN for every pixel, with it off (what the package sets) the gradients are finite and equal to eager.

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 GD_HIP_GRAPHS_FORCE=1 python tools/graph_replay_check.py   # -> finite False
    python tools/graph_replay_check.py                                                        # -> finite True
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd  # noqa: E402,F401  (sets the runtime flag unless the caller exported one)
import torch  # noqa: E402
import bench  # noqa: E402
from garmentdreamer_amd.gaussian_model import GaussianModel  # noqa: E402
from garmentdreamer_amd.guidance import sd21  # noqa: E402
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance  # noqa: E402
from garmentdreamer_amd.scene import synthetic_gaussians  # noqa: E402
from garmentdreamer_amd.sds_loop import SDSLoop  # noqa: E402

dev = torch.device("cuda", 0)
print("flag", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"))
# phase 1: what an earlier test / an earlier, smaller job leaves behind in the process
with torch.device(dev):
    unet = sd21.init_random_(sd21.UNet2DConditionModel(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4)))
    vae = sd21.init_random_(sd21.AutoencoderKLEncoder(block_out_channels=(64, 64, 128, 128)))
small = StableDiffusionGuidance({"use_hip_graphs": False, "grad_clip": [0, 1.5, 2.0, 1000], "guidance_scale": 1.0},
                                device=dev, unet=unet, vae=vae)
small.update_step(0, 0)
prompts = PromptEmbeddings.random(dev)
for it in range(2):
    g = torch.Generator(dev).manual_seed(it)
    rgb = torch.rand(2, 64, 64, 3, device=dev, generator=g).requires_grad_(True)
    out = small(rgb, prompts, torch.tensor([10.0, 20.0], device=dev), torch.tensor([0.0, 100.0], device=dev),
                torch.ones(2, device=dev) * 2, noise=torch.randn(2, 4, 64, 64, device=dev, generator=g),
                timesteps=torch.tensor([100 + it, 700], device=dev),
                vae_noise=torch.randn(2, 4, 64, 64, device=dev, generator=g))
    out["loss_sds"].backward()
# phase 2: the full-size loop with both graphs
V = 8
args = argparse.Namespace(views=V, gaussians=20000, res=512)
gm = GaussianModel.from_activated(synthetic_gaussians(20000, seed=0), device=dev)
guid = StableDiffusionGuidance({"guidance_scale": 100.0, "grad_clip": [0, 1.5, 2.0, 1000], "use_hip_graphs": True}, device=dev)
print("graphs in use:", guid.cfg.use_hip_graphs)
loop = SDSLoop(gm, guid, PromptEmbeddings.random(dev), torch.ones(3, device=dev))
gen = torch.Generator(device=dev)
ok = True
for s in range(3):
    gen.manual_seed(100 + s)
    noise = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    vn = torch.randn(V, 4, 64, 64, device=dev, generator=gen)
    t = torch.randint(20, 981, (V,), device=dev, generator=gen)
    o = loop.step(bench.camera_batch(args, s, list(range(V))), noise=noise, timesteps=t, vae_noise=vn)
    fin = bool(torch.isfinite(gm.flat_grad).all())
    ok &= fin
    print(f"step {s}: loss {o['loss'].item():.3f} gradients finite {fin}")
sys.exit(0 if ok else 1)
