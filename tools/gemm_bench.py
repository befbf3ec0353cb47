"""Own bf16 implicit-GEMM kernel used as nn.Linear (one tap) vs hipBLASLt on the transformer shapes (16 samples)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import garmentdreamer_amd  # noqa
import torch
import torch.nn.functional as F
from garmentdreamer_amd import nn_ops
DEV = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
g = torch.Generator(DEV).manual_seed(0)
tot0 = tot1 = 0
print("      M     K      N   hipBLASLt us   own us")
for M, K, N, cnt in [(65536, 320, 320, 15), (65536, 320, 960, 5), (65536, 320, 2560, 5), (65536, 1280, 320, 5),
                     (16384, 640, 640, 15), (16384, 640, 1920, 5), (16384, 640, 5120, 5), (16384, 2560, 640, 5),
                     (4096, 1280, 1280, 18), (4096, 1280, 3840, 6), (4096, 1280, 10240, 6), (4096, 5120, 1280, 6)]:
    x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g).to(torch.bfloat16)
    with torch.no_grad():
        ref = F.linear(x, w, b)
        got = nn_ops.linear(x, w, b)
        assert (ref.float() - got.float()).abs().max().item() <= 2e-2 * ref.float().abs().max().item() + 1e-2
        t0 = timeit(lambda: F.linear(x, w, b))
        t1 = timeit(lambda: nn_ops.linear(x, w, b))
    tot0 += t0 * cnt; tot1 += t1 * cnt
    print(f"{M:7d} {K:5d} {N:6d} {t0*1e6:10.1f} {t1*1e6:10.1f}")
print(f"weighted per UNet forward: hipBLASLt {tot0*1e3:.2f} ms, own {tot1*1e3:.2f} ms")
