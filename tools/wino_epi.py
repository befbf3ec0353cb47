#!/usr/bin/env python
"""Epilogue cost of the Winograd kernel: 128->128 @512 with / without bias, residual, statistics."""
import sys, time, torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops
N, ci, co, hw = 8, 128, 128, 512
if len(sys.argv) > 4:
    N, ci, co, hw = (int(a) for a in sys.argv[1:5])
cl = torch.channels_last
x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=cl)
b = torch.randn(co, device="cuda").to(torch.bfloat16)
r = torch.randn(N, co, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
rows = ((hw + 15) // 16) ** 2 * 8
part = torch.zeros(N * (co // 4) * rows * 2, dtype=torch.float32, device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    for name, fn in [("wino plain", lambda: nn_ops._wino_launch(x, w, None, None, co)),
                     ("wino +bias", lambda: nn_ops._wino_launch(x, w, b, None, co)),
                     ("wino +bias+res", lambda: nn_ops._wino_launch(x, w, b, r, co)),
                     ("wino +bias+res+stats", lambda: nn_ops._wino_launch(x, w, b, r, co, part)),
                     ("direct plain", lambda: nn_ops._patch_launch(x, w, None, None, co)),
                     ("direct +bias", lambda: nn_ops._patch_launch(x, w, b, None, co)),
                     ("direct +bias+res", lambda: nn_ops._patch_launch(x, w, b, r, co))]:
        print(f"{name:24s} {timeit(fn)*1e6:8.1f} us", flush=True)
