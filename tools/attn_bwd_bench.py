#!/usr/bin/env python
"""Own attention backward (csrc/nn_attention.hip, gd_nn_attention_d64_backward) against the library's flash backward on the
LoRA UNet's training shapes: time per call of the backward pass alone, and agreement of the two."""
import os
import sys
import torch
sys.path.insert(0, ".")
from garmentdreamer_amd import nn_ops  # noqa: E402

torch.manual_seed(0)
for (B, S, Skv, H) in ((1, 4096, 4096, 5), (1, 1024, 1024, 10), (1, 256, 256, 20), (1, 4096, 77, 5), (1, 1024, 77, 10), (2, 4096, 4096, 5)):
    C = H * 64
    q = torch.randn(B, S, H, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
    k = torch.randn(B, Skv, H, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
    v = torch.randn(B, Skv, H, 64, device="cuda").to(torch.bfloat16).requires_grad_(True)
    do = torch.randn(B, S, C, device="cuda").to(torch.bfloat16)
    res = {}
    for own in (1, 0):
        nn_ops._ATTN_BWD = bool(own)
        o = nn_ops.attention_d64_train(q, k, v)
        for _ in range(3):
            g = torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g = torch.autograd.grad(o, (q, k, v), do, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        res[own] = (e0.elapsed_time(e1) / 20 * 1e3, [t.float() for t in g])
    rel = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(res[1][1], res[0][1])]
    print(f"B {B} S {S} Skv {Skv} H {H}: own {res[1][0]:7.1f} us | library {res[0][0]:7.1f} us | {res[0][0] / res[1][0]:.2f}x | "
          f"max rel diff dq {rel[0]:.1e} dk {rel[1]:.1e} dv {rel[2]:.1e}")
