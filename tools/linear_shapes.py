"""Which library GEMMs the 8-view UNet forward (batch 16) still runs: shape, calls, time per call (CUDA events, eager)."""
import collections
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd.guidance import sd21

dev = "cuda:0"
torch.manual_seed(0)
unet = sd21.init_random_(sd21.UNet2DConditionModel()).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
for p in unet.parameters():
    p.requires_grad_(False)
B = 16
lat = torch.randn(B, 4, 64, 64, device=dev, dtype=torch.bfloat16)
t = torch.randint(20, 980, (B,), device=dev)
ctx = torch.randn(B, 77, 1024, device=dev, dtype=torch.bfloat16)
stats = collections.defaultdict(list)
real = {"linear": F.linear, "matmul": torch.matmul, "addmm": torch.addmm, "bmm": torch.bmm}


def wrap(name, fn):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        shapes = tuple(tuple(x.shape) for x in a if torch.is_tensor(x))
        stats[(name, shapes)].append((e0, e1))
        return out
    return f


F.linear = wrap("linear", real["linear"])
torch.matmul = wrap("matmul", real["matmul"])
torch.addmm = wrap("addmm", real["addmm"])
with torch.no_grad():
    for _ in range(2):
        unet(lat, t, ctx)
    stats.clear()
    unet(lat, t, ctx)
torch.cuda.synchronize()
rows = []
for (name, shapes), evs in stats.items():
    us = [a.elapsed_time(b) * 1e3 for a, b in evs]
    rows.append((sum(us), name, shapes, len(us), sum(us) / len(us)))
rows.sort(reverse=True)
print(f"library GEMM calls: {sum(r[3] for r in rows)}, total {sum(r[0] for r in rows) / 1e3:.2f} ms (eager, per UNet forward)")
for tot, name, shapes, n, avg in rows[:25]:
    print(f"{tot / 1e3:6.3f} ms  {n:3d} x {avg:7.1f} us  {name:7s} {shapes}")
