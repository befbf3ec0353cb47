#!/usr/bin/env python
"""One-launch GroupNorm(+SiLU) (gd_nn_groupnorm_silu_fused_forward) vs statistics + apply kernels on the UNet's shapes;
device time per call from a hipGraph of 20 calls.   python tools/gn_small_bench.py [N]"""
import sys
import torch
sys.path.insert(0, ".")
import tools.ablib  # noqa: F401,E402
from garmentdreamer_amd import nn_ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SH = [(320, 64), (640, 64), (960, 64), (320, 32), (640, 32), (960, 32), (1280, 32), (1920, 32), (640, 16), (1280, 16), (1920, 16),
      (2560, 16), (1280, 8), (2560, 8)]


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
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


for C, hw in SH:
    # a ring of tensors larger than the L2s, so consecutive calls do not find their input cached by the previous one
    xs = [torch.randn(N, C, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    w = torch.ones(C, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(C, device="cuda", dtype=torch.bfloat16)
    k = [0]

    def two():
        k[0] += 1
        return nn_ops._GroupNormSiLU.apply(xs[k[0] & 3], w, b, 32, 1e-5, True)

    def one():
        k[0] += 1
        return nn_ops._gn_fused_small(xs[k[0] & 3], w, b, 32, 1e-5, True)
    with torch.no_grad():
        t2 = graph_time(two)
        sup = nn_ops.lib().gd_nn_groupnorm_silu_fused_supported(N, hw * hw, C, 32)
        t1 = graph_time(one) if sup else float("nan")
    mb = N * C * hw * hw * 2 / 1e6
    print(f"N{N:2d} C{C:5d} @{hw:2d}: {mb:6.1f} MB  two-pass {t2:6.1f} us  one-launch {t1:6.1f} us  {t2 / t1 if sup else 0:4.2f}x"
          f"  ({2 * mb / t1 / 1e3 if sup else 0:4.2f} TB/s r+w)")
