#!/usr/bin/env python
"""The LoRA UNet's forward at batch 1, no-grad against training mode (eager): device time and launches per kernel family, to see what the
training forward pays over the inference one.   python tools/lora_fwd_modes.py"""
import collections
import re
import sys
import torch
sys.path.insert(0, ".")
import garmentdreamer_amd  # noqa: F401,E402
from garmentdreamer_amd.guidance import sd21  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda", 0)
with torch.device(dev):
    lora = sd21.init_random_(sd21.LoraUNet2DConditionModel(), 2)
lora = lora.to(torch.bfloat16).to(memory_format=torch.channels_last)
lora.trainables_to_fp32()
lora.freeze_base()
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(1, 4, 64, 64, device=dev, generator=g).to(torch.bfloat16)
ctx = torch.randn(1, 77, 1024, device=dev, generator=g).to(torch.bfloat16)
pose = torch.randn(1, 16, device=dev, generator=g)
t = torch.tensor([500.0], device=dev)


def fam(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if "Cijk" in n[:24]:
        return "library GEMM"
    if "at::native" in n or "rocclr" in n:
        m = re.search(r"(CUDAFunctor_\w+|\w+_kernel\w*|Cat\w+|copyBuffer|fillBuffer)", n)
        return "aten " + (m.group(1)[:28] if m else "other")
    return n.split("(")[0].split("<")[0][:40]


res = {}
for mode in ("no_grad", "train"):
    def run():
        if mode == "no_grad":
            with torch.no_grad():
                return lora(x, t, ctx, c=pose, shading="albedo")
        return lora(x, t, ctx, c=pose, shading="albedo")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        dt = getattr(e, "device_time_total", 0) or 0
        if e.device_type is not None and str(e.device_type).endswith("CUDA") and dt > 0:
            a = agg[fam(e.name)]
            a[0] += 1
            a[1] += dt
    res[mode] = agg
keys = sorted(set(res["no_grad"]) | set(res["train"]), key=lambda k: -(res["train"].get(k, [0, 0])[1]))
tn = sum(v[1] for v in res["no_grad"].values()); tt = sum(v[1] for v in res["train"].values())
print(f"no-grad forward {tn / 1e3:.2f} ms in {sum(v[0] for v in res['no_grad'].values())} launches | training forward {tt / 1e3:.2f} ms in {sum(v[0] for v in res['train'].values())} launches")
for k in keys[:40]:
    a, b = res["no_grad"].get(k, [0, 0.0]), res["train"].get(k, [0, 0.0])
    print(f"{k:42s} no-grad {a[0]:4d} x {a[1] / max(a[0], 1):6.1f} us = {a[1] / 1e3:6.3f} ms | train {b[0]:4d} x {b[1] / max(b[0], 1):6.1f} us = {b[1] / 1e3:6.3f} ms")
