#!/usr/bin/env python
"""GroupNorm-in-loader convolutions of the VAE encoder (conv(silu(GN(x))) + bias (+ residual), epilogue statistics):
direct patch kernel vs Winograd vs wide tile."""
import sys, time, torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SH = [(V, 128, 128, 512, 1), (V, 128, 128, 512, 0), (V, 128, 256, 256, 0), (V, 256, 256, 256, 1), (V, 256, 256, 256, 0)]
def timeit(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
cl = torch.channels_last
L = nn_ops.lib()
for (N, ci, co, hw, res) in SH:
    x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=cl)
    b = torch.randn(co, device="cuda").to(torch.bfloat16)
    r = torch.randn(N, co, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl) if res else None
    gw = torch.ones(ci, device="cuda", dtype=torch.bfloat16); gb = torch.zeros(ci, device="cuda", dtype=torch.bfloat16)
    mr = torch.tensor([0.0, 1.0], device="cuda").repeat(N * 32).contiguous()
    rows = (hw // 16) ** 2 * 8
    part = torch.zeros(N * (co // 4) * rows * 2, dtype=torch.float32, device="cuda")
    y = torch.empty(N, co, hw, hw, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=cl)
    st = torch.cuda.current_stream().cuda_stream
    rp = None if r is None else r.data_ptr()
    fd = lambda: L.gd_nn_conv3x3_gn_forward_stats(st, x.data_ptr(), mr.data_ptr(), gw.data_ptr(), gb.data_ptr(), 32, 1, w.data_ptr(),
                                                  b.data_ptr(), 0, rp, y.data_ptr(), N, hw, hw, ci, co, part.data_ptr())
    fw = lambda: nn_ops._wino_gn_launch(x, mr, gw, gb, 32, True, w, b, r, co, part)
    fz = lambda: nn_ops._wide_launch(x, w, b, r, co, part, gn=(mr, gw, gb, 32, True))
    with torch.no_grad():
        td, tw, tz = timeit(fd), timeit(fw), timeit(fz)
        td, tw, tz = min(td, timeit(fd)), min(tw, timeit(fw)), min(tz, timeit(fz))
    print(f"N{N} {ci}->{co} @{hw} res={res}: direct {td*1e6:7.1f}us | wino {tw*1e6:7.1f}us {td/tw:4.2f}x | wide {tz*1e6:7.1f}us {td/tz:4.2f}x", flush=True)
