#!/usr/bin/env python
"""Where the step's remaining aten / runtime-copy launches come from: one eager VSD iteration (or one 1-view SDS guidance step with
--sds) under torch.profiler, every op with device time that is NOT an own HIP kernel or a library GEMM grouped by (op, input
shapes, innermost frame inside this package; "<autograd>" for ops the backward engine runs).   python tools/aten_sites.py [--sds]"""
import collections
import os
import sys
import torch
sys.path.insert(0, ".")
import garmentdreamer_amd  # noqa: F401,E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda", 0)
sds = "--sds" in sys.argv
from garmentdreamer_amd.guidance import sd21  # noqa: E402
if sds:
    # the guidance half of a 1-view SDS step: VAE encode with gradient + the frozen UNet on 2 latents (classifier-free guidance)
    from garmentdreamer_amd.guidance.sd_vsd import StableDiffusionVSD
    gd = StableDiffusionVSD(dev, fp16=True, use_hip_graphs=False)
    g = torch.Generator(device=dev).manual_seed(7)
    emb = torch.randn(2, 77, 1024, device=dev, generator=g)
    img = torch.rand(1, 3, 512, 512, device=dev, generator=g, requires_grad=True)

    def step():
        lat = gd.encode_imgs(img)
        with torch.no_grad():
            eps = gd._frozen_unet(torch.cat([lat.detach()] * 2).to(torch.bfloat16), torch.tensor([500, 500], device=dev), emb.to(torch.bfloat16))
        img.grad = None
        (lat * eps[:1].float()).sum().backward()
else:
    from garmentdreamer_amd.guidance.sd_vsd import LoraUnet, StableDiffusionVSD
    from garmentdreamer_amd.flat_adam import FlatAdam
    gd = StableDiffusionVSD(dev, fp16=True, use_hip_graphs=False)
    with torch.device(dev):
        lora = sd21.init_random_(sd21.LoraUNet2DConditionModel(), 2)
    lora = lora.to(torch.bfloat16).to(memory_format=torch.channels_last)
    lora.trainables_to_fp32()
    train = lora.freeze_base()
    q = LoraUnet(lora)
    opt = FlatAdam.for_lora_unet(lora, train, lr=1e-4)
    g = torch.Generator(device=dev).manual_seed(7)
    gd.set_text_embeds(torch.randn(1, 77, 1024, device=dev, generator=g), torch.randn(1, 77, 1024, device=dev, generator=g))
    img = torch.rand(1, 3, 512, 512, device=dev, generator=g, requires_grad=True)

    def step():
        pose = torch.randn(1, 16, device=dev, generator=g)
        loss, _, latents = gd.train_step(img, guidance_scale=7.5, q_unet=q, pose=pose, shading="albedo")
        img.grad = None
        loss.backward()
        lu = gd.lora_train_loss(q, latents, pose, shading="albedo", unet_bs=1)
        opt.zero_grad(set_to_none=True)
        lu.backward()
        opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

pkg = os.path.abspath("garmentdreamer_amd")
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0 or not e.name.startswith("aten::"):
        continue
    if e.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::matmul", "aten::linear"):
        continue
    site = "<autograd>"
    for fr in (e.stack or []):
        if "garmentdreamer_amd" in fr or "bench.py" in fr or "tools/" in fr:
            site = fr.replace(pkg + "/", "")[:110]
            break
    shapes = str([s for s in (e.input_shapes or []) if s])[:90]
    k = (e.name, shapes, site)
    rows[k][0] += 1
    rows[k][1] += dt
tot_n = sum(v[0] for v in rows.values())
tot_t = sum(v[1] for v in rows.values())
print(f"{'VSD iteration' if not sds else '1-view SDS guidance step'}: {tot_n} aten ops with device time, {tot_t / 1e3:.2f} ms (GEMMs excluded)")
for (name, shapes, site), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{n:4d} x {t / max(n, 1):6.1f} us = {t / 1e3:6.3f} ms  {name:28s} {shapes:90s} {site}")
