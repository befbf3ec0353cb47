#!/usr/bin/env python
"""One convolution shape, one kernel, a few launches (for rocprofv3 / PMC runs): wino_one.py N Cin Cout HW wino|direct [iters]"""
import sys
import torch
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops
N, ci, co, hw = (int(a) for a in sys.argv[1:5])
which = sys.argv[5]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
cl = torch.channels_last
torch.manual_seed(0)
x = torch.randn(N, ci, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=cl)
w = (torch.randn(co, ci, 3, 3, device="cuda") / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=cl)
b = torch.randn(co, device="cuda").to(torch.bfloat16)
fn = nn_ops._wino_launch if which == "wino" else nn_ops._patch_launch
with torch.no_grad():
    for _ in range(iters):
        y = fn(x, w, b, None, co)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
