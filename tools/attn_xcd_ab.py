#!/usr/bin/env python
"""Self-attention forward (head_dim 64) with the query tiles of a head on ONE XCD against round-robin over the eight: device time per call
(hipGraph of 10 calls) on the UNet's shapes.  GD_NN_ATTN_XCD is read at the first launch: run once per value.
   GD_NN_ATTN_XCD=0 python tools/attn_xcd_ab.py ; GD_NN_ATTN_XCD=1 python tools/attn_xcd_ab.py"""
import os
import sys
import torch
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402   (GD_NN_LIB=ablate/libgd_nn_<name>.so: timing builds of tools/attn_ablate.sh)
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


print("GD_NN_ATTN_XCD =", os.environ.get("GD_NN_ATTN_XCD", "1"))
for B, H, S in ((16, 5, 4096), (16, 10, 1024), (16, 20, 256), (2, 5, 4096), (2, 10, 1024), (1, 5, 4096))[:int(os.environ.get("ATTN_SHAPES", "6"))]:
    q = torch.randn(B, S, H, 64, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, S, H, 64, device="cuda").to(torch.bfloat16)
    vt = torch.randn(B, H * 64, S, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        t = graph_time(lambda: nn_ops.attention_d64_vt(q, k, vt))
    fl = 4.0 * B * H * S * S * 64
    print(f"B{B} H{H} S{S}: {t:7.1f} us  {fl / t / 1e6:6.0f} TFLOP/s")
