#!/usr/bin/env python
"""Repeat one 3x3 convolution on identical inputs and report where (GEMM row, channel) the bits differ between runs."""
import sys

import torch

sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
from garmentdreamer_amd import nn_ops  # noqa: E402

N, C, H, W, Co = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 1280, 8, 8, 1280)))
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, C, H, W, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(Co, C, 3, 3, generator=g) * 0.02).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)
b = torch.randn(Co, generator=g).to(dev, torch.bfloat16)
L = nn_ops.lib()
for split in (-1, 1, 2, 3, 5, 9):
    L.gd_nn_conv_force_split(split)
    for sync in (True, False):
        ref = nn_ops.conv3x3(x, w, b, None)
        torch.cuda.synchronize()
        outs = []
        for _ in range(20):
            outs.append(nn_ops.conv3x3(x, w, b, None))
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        nd = [int((o != ref).sum()) for o in outs]
        line = f"split {split:2d} sync {int(sync)}: runs differing {sum(1 for v in nd if v)} / 20, elements {max(nd)}"
        if max(nd):
            o = outs[max(range(20), key=lambda i: nd[i])]
            idx = (o != ref).permute(0, 2, 3, 1).reshape(-1, Co).nonzero()
            rows, cols = idx[:, 0], idx[:, 1]
            line += (f"; rows {int(rows.min())}..{int(rows.max())} ({rows.unique().numel()} distinct), channels "
                     f"{int(cols.min())}..{int(cols.max())} ({cols.unique().numel()} distinct), max|d| "
                     f"{(o.float() - ref.float()).abs().max().item():.3e}")
        print(line)
L.gd_nn_conv_force_split(-1)
