#!/usr/bin/env python
"""Correctness of the Winograd convolution kernels (plain and GroupNorm-in-loader) against fp32 PyTorch, on the cases
the direct kernels' tests cover: ragged sizes, per-image bias, residual, Cout not a multiple of 128, one chunk, epilogue
GroupNorm statistics.  Prints max error / max|ref| per case; exit code 1 if any case is above the direct kernels' bar."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops

WIDE = len(sys.argv) > 1 and sys.argv[1] == "wide"      # the wide-tile direct kernel (csrc/nn_conv_wide.h) instead
torch.manual_seed(1)
dev = "cuda"
cl = torch.channels_last
CASES = [  # N, Cin, Cout, H, W, per-image bias, residual, gn
    (1, 32, 64, 16, 16, False, False, False), (2, 64, 128, 48, 32, True, True, False), (3, 96, 72, 40, 56, False, True, False),
    (2, 128, 320, 64, 64, True, False, False), (1, 128, 128, 100, 36, False, False, False), (8, 128, 128, 128, 128, False, True, False),
    (2, 64, 128, 48, 32, True, True, True), (1, 32, 64, 16, 16, False, False, True), (2, 128, 128, 64, 80, False, True, True),
    (3, 320, 320, 32, 32, True, False, True), (2, 256, 256, 72, 40, False, False, True), (5, 128, 256, 32, 48, False, False, False),
    (40, 64, 64, 32, 32, False, False, False), (40, 64, 64, 32, 32, False, True, True),
]
bad = 0
for (N, ci, co, H, W, pib, res, gn) in CASES:
    x = (torch.randn(N, ci, H, W, device=dev) * 1.5 + 0.3).to(torch.bfloat16).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)).to(torch.bfloat16).contiguous(memory_format=cl)
    b = torch.randn(N, co, device=dev).to(torch.bfloat16) if pib else torch.randn(co, device=dev).to(torch.bfloat16)
    r = torch.randn(N, co, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=cl) if res else None
    groups = 32
    rows = ((H + 15) // 16) * ((W + 15) // 16) * 8
    part = torch.full((N * (co // 4) * rows * 2,), float("nan"), dtype=torch.float32, device=dev)
    with torch.no_grad():
        xin = x.float()
        if gn:
            gw = (torch.randn(ci, device=dev) * 0.5 + 1).to(torch.bfloat16)
            gb = (torch.randn(ci, device=dev) * 0.5).to(torch.bfloat16)
            xin = F.silu(F.group_norm(xin, groups, gw.float(), gb.float(), 1e-6))
            xg = x.float().reshape(N, groups, -1)
            mr = torch.stack([xg.mean(-1), (xg.var(-1, unbiased=False) + 1e-6).rsqrt()], -1).reshape(-1).contiguous()
        ref = F.conv2d(xin, w.float(), None, padding=1)
        ref = ref + (b.float()[:, :, None, None] if pib else b.float()[None, :, None, None])
        if res:
            ref = ref + r.float()
        if WIDE:
            y = nn_ops._wide_launch(x, w, b, r, co, part, gn=(mr, gw, gb, groups, True) if gn else None)
        elif gn:
            y = nn_ops._wino_gn_launch(x, mr, gw, gb, groups, True, w, b, r, co, part)
        else:
            y = nn_ops._wino_launch(x, w, b, r, co, part)
        torch.cuda.synchronize()
        err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
        # epilogue statistics: per (image, channel quad) sum / sum of squares of the STORED values
        pp = part.view(N, co // 4, rows, 2).double().sum(2)
        yq = y.float().double().view(N, co // 4, 4, H * W)
        s_ref = torch.stack([yq.sum((2, 3)), (yq * yq).sum((2, 3))], -1)
        serr = ((pp - s_ref).abs().max() / s_ref.abs().max()).item()
        ok = err < (6e-3 if WIDE else 1.2e-2) and serr < 1e-4 and bool(torch.isfinite(part).all())
        bad += not ok
        print(f"N{N} {ci}->{co} {H}x{W} pib={int(pib)} res={int(res)} gn={int(gn)}: err {err:.2e}  stats err {serr:.1e}  {'ok' if ok else 'FAIL'}",
              flush=True)
sys.exit(1 if bad else 0)
