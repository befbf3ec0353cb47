"""Time of the VAE encoder's first convolution (8 x 3 x 512^2 -> 128 channels) with / without the epilogue statistics."""
import torch
from garmentdreamer_amd import nn_ops

dev = "cuda:0"
cl = torch.channels_last
x = torch.rand(8, 3, 512, 512, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
w = (torch.randn(128, 3, 3, 3, device=dev) / 4).to(torch.bfloat16).contiguous(memory_format=cl)
b = torch.randn(128, device=dev).to(torch.bfloat16)
for nn_ in (None, (32, 1e-6)):
    for _ in range(3):
        nn_ops._ConvSmallCin.apply(x, w, b, nn_)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        nn_ops._ConvSmallCin.apply(x, w, b, nn_)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"first conv, statistics {'on ' if nn_ else 'off'}: {us:7.1f} us  ({8 * 512 * 512 * 128 * 2 / us / 1e6:.2f} TB/s of output)")
