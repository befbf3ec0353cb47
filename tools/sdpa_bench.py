import time
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for (B, H, S, D, Skv) in [(16, 5, 4096, 64, 4096), (16, 10, 1024, 64, 1024), (16, 20, 256, 64, 256), (16, 5, 4096, 64, 77),
                          (2, 5, 4096, 64, 4096)]:
    q = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, H, Skv, D, device="cuda", dtype=torch.bfloat16)
    fl = 4.0 * B * H * S * Skv * D
    res = []
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        try:
            with sdpa_kernel(be), torch.no_grad():
                t = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
            res.append(f"{name} {t*1e6:7.1f}us {fl/t/1e12:5.0f}TF")
        except Exception as e:
            res.append(f"{name} n/a ({str(e)[:30]})")
    # layout variant: [B,S,H,D] strided (what the model passes)
    q2 = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)
    k2 = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)
    v2 = torch.randn(B, Skv, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)
    with torch.no_grad():
        t = timeit(lambda: F.scaled_dot_product_attention(q2, k2, v2))
    res.append(f"default/BSHD {t*1e6:7.1f}us {fl/t/1e12:5.0f}TF")
    print(f"B{B} H{H} S{S} Skv{Skv}: " + " | ".join(res))
