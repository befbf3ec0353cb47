"""Small-M transformer linears (1 view per GPU: batch 2; the VSD iteration: batch 1): hipBLASLt vs the own one-tap implicit-GEMM
kernel (nn_ops.linear).  These products are latency-bound (a few microseconds of arithmetic); times are per call from
back-to-back launches inside one hipGraph replay (what the guidance step does)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd  # noqa
import torch
import torch.nn.functional as F
from garmentdreamer_amd import nn_ops
DEV = "cuda:0"
g = torch.Generator(DEV).manual_seed(0)


def graph_time(fn, reps=40):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / reps * 1e3


print("      M     K      N   hipBLASLt us   own us")
levels = [(320, 4096), (640, 1024), (1280, 256), (1280, 64)]
for batch in (1, 2):
    for C, tok in levels:
        M = batch * tok
        for K, N in ((C, C), (C, 3 * C), (C, 8 * C), (4 * C, C), (1024, 2 * C)):
            Mx = M if K != 1024 else batch * 77
            x = torch.randn(Mx, K, device=DEV, generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device=DEV, generator=g).to(torch.bfloat16)
            with torch.no_grad():
                ref = F.linear(x, w, b)
                got = nn_ops.linear(x, w, b)
                assert (ref.float() - got.float()).abs().max().item() <= 2e-2 * ref.float().abs().max().item() + 1e-2
                t0 = graph_time(lambda: F.linear(x, w, b))
                t1 = graph_time(lambda: nn_ops.linear(x, w, b))
            print(f"{Mx:7d} {K:5d} {N:6d} {t0:10.1f} {t1:10.1f}  {'OWN' if t1 < 0.95 * t0 else ''}")
