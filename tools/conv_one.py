#!/usr/bin/env python
"""Run the MFMA conv kernel a few times on one shape (for rocprofv3 --pmc).  args: N Cin Cout HW variant"""
import sys
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops
N, ci, co, hw, v = [int(a) for a in sys.argv[1:6]]
x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
b = torch.randn(co, device="cuda").to(torch.bfloat16)
nn_ops.lib().gd_nn_conv_force_variant(v)
with torch.no_grad():
    for _ in range(4):
        nn_ops.conv3x3(x, w, b)
torch.cuda.synchronize()
