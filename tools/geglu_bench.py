import sys, time, torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops
import torch.nn.functional as F
x = (torch.randn(65536, 2560, device="cuda") * 2).to(torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    y = nn_ops.geglu(x); h, g = x.chunk(2, -1); e = h * F.gelu(g)
    print("geglu", timeit(lambda: nn_ops.geglu(x)) * 1e6, "us; mismatch frac", (y != e).float().mean().item(), "max diff", (y.float() - e.float()).abs().max().item())
