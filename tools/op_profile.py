#!/usr/bin/env python
"""Attribute GPU kernel time of one eager SDS step to aten ops + input shapes (torch.profiler), to find
the elementwise / copy passes worth fusing.  python tools/op_profile.py [--views 8] > gpurun_out/ops.txt"""
import argparse
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
import bench  # noqa: E402
from garmentdreamer_amd import _native  # noqa: E402
from garmentdreamer_amd.guidance.stable_diffusion_guidance import PromptEmbeddings, StableDiffusionGuidance  # noqa
from garmentdreamer_amd.scene import GaussianParams, synthetic_gaussians  # noqa: E402
from garmentdreamer_amd.sds_loop import SDSLoop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--top", type=int, default=70)
a = ap.parse_args()
args = argparse.Namespace(views=a.views, gaussians=100000, res=512)
device = torch.device("cuda", 0)
_native.lib()
scene = synthetic_gaussians(args.gaussians, seed=0, sh_degree=0)
gaussians = GaussianParams(scene, sh_degree=0, device=device)
guidance = StableDiffusionGuidance({"guidance_scale": 100.0, "grad_clip": [0, 1.5, 2.0, 1000], "use_hip_graphs": False},
                                   device=device)
loop = SDSLoop(gaussians, guidance, PromptEmbeddings.random(device), torch.ones(3, device=device))
gen = torch.Generator(device=device)
view_ids = list(range(args.views))


def one_step(step):
    batch = bench.camera_batch(args, step, view_ids)
    V = args.views
    gen.manual_seed(1234 + step)
    noise = torch.randn(V, 4, 64, 64, device=device, generator=gen)
    vae_noise = torch.randn(V, 4, 64, 64, device=device, generator=gen)
    t = torch.randint(20, 981, (V,), device=device, generator=gen)
    loop.step(batch, noise=noise, timesteps=t, vae_noise=vae_noise)


for s in range(2):
    one_step(s)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    one_step(2)
    torch.cuda.synchronize()

# kernel -> launching op: walk events, use the op's own (self) device time
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0 or e.device_type.name != "CPU":
        continue
    shapes = str(e.input_shapes)[:110] if e.input_shapes else ""
    agg[(e.name, shapes)][0] += dt
    agg[(e.name, shapes)][1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
total = sum(v[0] for v in agg.values())
print(f"total self device time {total/1e3:.2f} ms")
byname = collections.defaultdict(float)
for (n, _), v in agg.items():
    byname[n] += v[0]
print("---- by op")
for n, t in sorted(byname.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{t/1e3:8.3f} ms  {n}")
print("---- by op + shapes")
for (n, sh), (t, c) in rows[:a.top]:
    print(f"{t/1e3:8.3f} ms x{c:4d}  {n:40s} {sh}")
