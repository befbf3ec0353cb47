#!/usr/bin/env python
"""The VAE encoder's conv_out (512 -> 8 channels at 64^2) forward and input gradient per call (hipGraph of 10 calls).
   python tools/conv_out_bench.py"""
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


for N in (8, 1):
    x = torch.randn(N, 512, 64, 64, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(8, 512, 3, 3, device="cuda") / 70).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.randn(8, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, 8, 64, 64, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    class Ctx:
        weight = w
        has_res = False
        needs_input_grad = (True, False, False, False)
        bias_meta = (1, torch.bfloat16)
    with torch.no_grad():
        tf = graph_time(lambda: nn_ops._conv_launch(x, w, b, None, 8))
        tb = graph_time(lambda: nn_ops._Conv3x3.backward(Ctx, dy))
    print(f"N{N}: conv_out forward {tf:6.1f} us ({x.numel() * 2 / tf / 1e6:.2f} TB/s of x) | input gradient {tb:6.1f} us ({x.numel() * 2 / tb / 1e6:.2f} TB/s of dx)")
