#!/usr/bin/env python
"""Input gradient of the VAE's first convolution (3 <- 128 channels): the dy-read-once kernel (csrc/nn_conv_first_dgrad.h) against the
padded implicit-GEMM form, device time per call from a hipGraph of 10 calls.   python tools/first_dgrad_bench.py"""
import sys
import torch
sys.path.insert(0, ".")
import garmentdreamer_amd  # noqa: F401,E402
from garmentdreamer_amd import nn_ops  # noqa: E402


def graph_time(fn, reps=10):
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


for N, H in ((8, 512), (1, 512), (4, 512)):
    w = (torch.randn(128, 3, 3, 3, device="cuda") / 3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, 128, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    class Ctx:
        weight = w
        needs_input_grad = (True,)
    t = {}
    for new in (True, False):
        nn_ops._FIRST_DGRAD = new
        t[new] = graph_time(lambda: nn_ops._ConvSmallCin.backward(Ctx, dy))
    gb = dy.numel() * 2 / 1e9
    print(f"N{N} @{H}^2: padded implicit GEMM {t[False]:7.1f} us | read-once {t[True]:7.1f} us ({gb / t[True] * 1e3:5.2f} TB/s of dy)  {t[False] / t[True]:.2f}x")
