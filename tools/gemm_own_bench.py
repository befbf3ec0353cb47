#!/usr/bin/env python
"""The own GEMM (csrc/nn_gemm.hip: persistent 256 x 256 x 64 tiles, ten-slot LDS-DMA ring) against hipBLASLt on the transformer /
VAE linears of the 8-view step, per call inside a hipGraph of 20 calls (the round-5 table's method, profiles/r05_gemm_shapes.txt),
plus the fused forms against what they replace:  GEGLU projection + geglu_kernel  vs  gemm_geglu;  linear + add  vs  gemm(residual).
    python tools/gemm_own_bench.py [views=8]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
import garmentdreamer_amd  # noqa: F401
from garmentdreamer_amd import nn_ops

V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 2 * V


def graph_time(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * reps)


levels = [(320, 4096), (640, 1024), (1280, 256), (1280, 64)]
shapes = [(N * tok, K, Nn) for C, tok in levels for K, Nn in ((C, C), (C, 2 * C), (C, 8 * C), (4 * C, C))]
shapes += [(N * 77, 1024, 24960), (V * 4096, 512, 1536), (V * 4096, 512, 512), (V * 16384, 256, 512)]
print(f"# own GEMM vs hipBLASLt, {V} views (UNet batch {N}); us per call in a hipGraph of 20, TFLOP/s in brackets")
tot_lib = tot_own = 0.0
for Mx, K, Nn in shapes:
    x = torch.randn(Mx, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Nn, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(Nn, device="cuda").to(torch.bfloat16)
    fl = 2.0 * Mx * K * Nn
    with torch.no_grad():
        ref = F.linear(x.float(), w.float(), b.float())
        t_lib = graph_time(lambda: F.linear(x, w, b))
        if not nn_ops.gemm_supported(x, w, b):
            print(f"M{Mx:6d} K{K:5d} N{Nn:6d}: hipBLASLt {t_lib:7.1f}us  own: unsupported")
            continue
        t_own = graph_time(lambda: nn_ops.gemm(x, w, b))
        t_lib = min(t_lib, graph_time(lambda: F.linear(x, w, b)))       # interleaved, best of two each
        t_own = min(t_own, graph_time(lambda: nn_ops.gemm(x, w, b)))
        err = (nn_ops.gemm(x, w, b).float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"M{Mx:6d} K{K:5d} N{Nn:6d}: hipBLASLt {t_lib:7.1f}us [{fl / t_lib / 1e6:5.0f}]  own {t_own:7.1f}us [{fl / t_own / 1e6:5.0f}]  "
          f"own/lib {t_lib / t_own:5.2f}x  err {err:.1e}", flush=True)

print("# fused epilogues against the pairs they replace")
for C, tok in levels[1:]:
    Mx = N * tok
    x = torch.randn(Mx, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(8 * C, C, device="cuda") / C ** 0.5).to(torch.bfloat16)
    b = torch.randn(8 * C, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        t_pair = graph_time(lambda: nn_ops.geglu(F.linear(x, w, b)))
        t_f = graph_time(lambda: nn_ops.gemm_geglu(x, w, b))
        a, bb = nn_ops.geglu(nn_ops.gemm(x, w, b)), nn_ops.gemm_geglu(x, w, b)
        same = torch.equal(a, bb)
        h, g = F.linear(x.float(), w.float(), b.float()).chunk(2, -1)
        ref = h * F.gelu(g)
        err = (bb.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"GEGLU M{Mx:6d} K{C:5d} inner {4 * C:5d}: hipBLASLt + geglu {t_pair:7.1f}us   own fused {t_f:7.1f}us  {t_pair / t_f:5.2f}x  "
          f"bits == own gemm + geglu: {same}  err {err:.1e}", flush=True)
    # output projection + residual
    x4 = torch.randn(Mx, 4 * C, device="cuda").to(torch.bfloat16)
    w4 = (torch.randn(C, 4 * C, device="cuda") / (4 * C) ** 0.5).to(torch.bfloat16)
    b4 = torch.randn(C, device="cuda").to(torch.bfloat16)
    r = torch.randn(Mx, C, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        t_pair = graph_time(lambda: F.linear(x4, w4, b4) + r)
        t_addmm = graph_time(lambda: torch.addmm(r + b4, x4, w4.t()))
        t_f = graph_time(lambda: nn_ops.gemm(x4, w4, b4, r))
    print(f"FF out + residual M{Mx:6d} K{4 * C:5d} N{C:5d}: linear + add {t_pair:7.1f}us  addmm {t_addmm:7.1f}us  own fused {t_f:7.1f}us",
          flush=True)
