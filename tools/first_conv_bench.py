import sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops
cl = torch.channels_last
x = torch.rand(8, 3, 512, 512, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
w = (torch.randn(128, 3, 3, 3, device="cuda") / 5).to(torch.bfloat16).contiguous(memory_format=cl)
b = torch.randn(128, device="cuda").to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    print("own", timeit(lambda: nn_ops.conv3x3_small_cin(x, w, b)) * 1e6, "us; torch", timeit(lambda: F.conv2d(x, w, b, padding=1)) * 1e6, "us")
# input gradient of the first conv: MFMA kernel on the flipped weights padded to 4 output channels
xg = x.clone().requires_grad_(True)
y = nn_ops.conv3x3_small_cin(xg, w, b)
gy = torch.randn_like(y)
def bwd():
    xg.grad = None
    y.backward(gy, retain_graph=True)
print("dgrad (incl. autograd overhead)", timeit(bwd) * 1e6, "us")
ref = torch.nn.grad.conv2d_input(x.shape, w.float(), gy.float(), padding=1)
bwd()
print("dgrad max err vs fp32", (xg.grad.float() - ref).abs().max().item(), "scale", ref.abs().max().item())
