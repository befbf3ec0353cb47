#!/usr/bin/env python
"""Own head_dim-64 attention forward vs torch SDPA on the UNet's self-attention shapes."""
import sys
import time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import ablib  # noqa: F401,E402  (GD_NN_LIB / GD_RASTER_LIB -> use_library)
from garmentdreamer_amd import nn_ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for (B, H, S) in [(16, 5, 4096), (16, 10, 1024), (16, 20, 256), (16, 20, 64), (2, 5, 4096), (2, 10, 1024)]:
    qkv = [torch.randn(B, S, H * 64, device="cuda").to(torch.bfloat16) for _ in range(3)]
    q, k, v = [t.view(B, S, H, 64) for t in qkv]
    with torch.no_grad():
        ref = F.scaled_dot_product_attention(q.transpose(1, 2).float(), k.transpose(1, 2).float(), v.transpose(1, 2).float())
        ref = ref.transpose(1, 2).reshape(B, S, H * 64)
        out = nn_ops.attention_d64(q, k, v)
        err = (out.float() - ref).abs().max().item()
        t_own = timeit(lambda: nn_ops.attention_d64(q, k, v))
        t_sd = timeit(lambda: F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
                      .transpose(1, 2).reshape(B, S, -1))
    fl = 4.0 * B * H * S * S * 64
    print(f"B{B} H{H} S{S}: own {t_own*1e6:7.1f}us {fl/t_own/1e12:5.0f}TF | sdpa {t_sd*1e6:7.1f}us {fl/t_sd/1e12:5.0f}TF | max err {err:.4f} (ref max {ref.abs().max().item():.2f})")
