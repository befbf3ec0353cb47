"""Kernels of the VAE mid-block attention (8 images, 64x64 tokens, 512 channels), forward + backward, by autograd op."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd, torch
from torch.profiler import profile, ProfilerActivity
from garmentdreamer_amd.guidance import sd21
dev = "cuda:0"
att = sd21._VAEAttention(512).to(dev).to(torch.bfloat16).requires_grad_(False)
x = torch.randn(8, 512, 64, 64, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
gy = torch.randn_like(x)
for _ in range(2):
    x.grad = None
    att(x).backward(gy)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    x.grad = None
    att(x).backward(gy)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in prof.key_averages()]
rows = [r for r in rows if r[2] > 0]
rows.sort(key=lambda r: -r[2])
for k, c, t in rows[:40]:
    print(f"{t:9.1f} us x{c:3d}  {k[:110]}")
